// Element kernels of the IPCS path, row-centric ("owner computes"): one lane owns one matrix
// row / vector entry and walks the cells adjacent to its dof, so nothing is scattered with
// atomics, every sum has a fixed order (bit-reproducible) and all matrix traffic is
// coalesced in the SELL-64 layout.  Quadrature tables are compile-time constants (fe_tables.h,
// degree-5 rules: every integrand of the path is a polynomial of degree <= 5 on affine cells,
// so the rule is exact, as FFCx's is in the reference).
#include "fe_tables.h"
#include "fe_tables_h.h"
#include "fe_tables_h3.h"
#include "ox_kernels.h"
#include <stdlib.h>
#include <algorithm>

#define OX_KIND_MASS 0
#define OX_KIND_STIFF 1
#define OX_KIND_CONV 2

// RULE 0: the degree-5 rules (every form of a P1 / P2 velocity is integrated exactly); RULE 1: the degree-9 rule a P3
// velocity needs (convection: 3 + 2 + 3 = 8) -- 25 collapsed Gauss-Jacobi points on triangles, 70 Grundmann-Moeller points
// on tetrahedra (round 5).  A kernel that couples two elements tabulates both on the rule of the higher one (OX_RULE_OF).
// P3 = Basix's gll_warped variant (reference fracstep.py:170,181).
#define OX_RULE_OF(DA, DB) (((DA) == 3 || (DB) == 3) ? 1 : 0)
template <int GDIM, int DEG, int RULE = (DEG == 3 ? 1 : 0)>
struct Elem {
  static_assert(DEG >= 1 && DEG <= 3, "Lagrange degree 1, 2, 3");
  static_assert(DEG < 3 || RULE == 1, "P3 needs the degree-9 rule");
  static constexpr int NV = GDIM + 1;
  // P3: vertices + 2 per edge + the cell's own dof (triangles) / one per face (tetrahedra)
  static constexpr int ND = DEG == 1 ? GDIM + 1 : (DEG == 3 ? (GDIM == 2 ? 10 : 20) : (GDIM == 2 ? 6 : 10));
  static constexpr int NQ = RULE == 1 ? (GDIM == 2 ? OX_NQ2H : OX_NQ3H) : (GDIM == 2 ? OX_NQ2 : OX_NQ3);
  static constexpr int GS = GDIM == 2 ? 6 : 10;
  __host__ __device__ static constexpr double w(int q) {
    if constexpr (RULE == 1 && GDIM == 3) return OX_QW3H[q];
    else if constexpr (RULE == 1) return OX_QW2H[q];
    else return GDIM == 2 ? OX_QW2[q] : OX_QW3[q];
  }
  __host__ __device__ static constexpr double phi(int q, int k) {
    if constexpr (RULE == 1 && GDIM == 3 && DEG == 1) return OX_PHI3H_1[q][k];
    else if constexpr (RULE == 1 && GDIM == 3 && DEG == 2) return OX_PHI3H_2[q][k];
    else if constexpr (RULE == 1 && GDIM == 3) return OX_PHI3H_3[q][k];
    else if constexpr (RULE == 1 && DEG == 1) return OX_PHI2H_1[q][k];
    else if constexpr (RULE == 1 && DEG == 2) return OX_PHI2H_2[q][k];
    else if constexpr (RULE == 1) return OX_PHI2H_3[q][k];
    else if constexpr (GDIM == 2 && DEG == 1) return OX_PHI2_1[q][k];
    else if constexpr (GDIM == 2 && DEG == 2) return OX_PHI2_2[q][k];
    else if constexpr (GDIM == 3 && DEG == 1) return OX_PHI3_1[q][k];
    else return OX_PHI3_2[q][k];
  }
  __host__ __device__ static constexpr double dphi(int q, int k, int b) {
    if constexpr (RULE == 1 && GDIM == 3 && DEG == 1) return OX_DPHI3H_1[q][k][b];
    else if constexpr (RULE == 1 && GDIM == 3 && DEG == 2) return OX_DPHI3H_2[q][k][b];
    else if constexpr (RULE == 1 && GDIM == 3) return OX_DPHI3H_3[q][k][b];
    else if constexpr (RULE == 1 && DEG == 1) return OX_DPHI2H_1[q][k][b];
    else if constexpr (RULE == 1 && DEG == 2) return OX_DPHI2H_2[q][k][b];
    else if constexpr (RULE == 1) return OX_DPHI2H_3[q][k][b];
    else if constexpr (GDIM == 2 && DEG == 1) return OX_DPHI2_1[q][k][b];
    else if constexpr (GDIM == 2 && DEG == 2) return OX_DPHI2_2[q][k][b];
    else if constexpr (GDIM == 3 && DEG == 1) return OX_DPHI3_1[q][k][b];
    else return OX_DPHI3_2[q][k][b];
  }
};

// Tables indexed by a RUNTIME local dof index (the lane's own row dof): device-resident
// copies, w_q*phi_i(q) and dphi_i(q,b).
template <int GDIM, int DEG, int RULE = (DEG == 3 ? 1 : 0)>
struct RtTab {
  using E = Elem<GDIM, DEG, RULE>;
  double wphi[E::ND][E::NQ];
  double dphi[E::ND][E::NQ][GDIM + 1];
  double iphi[E::ND];  // int_ref phi_i
};
template <int GDIM, int DEG, int RULE = (DEG == 3 ? 1 : 0)>
constexpr RtTab<GDIM, DEG, RULE> make_rt() {
  using E = Elem<GDIM, DEG, RULE>;
  RtTab<GDIM, DEG, RULE> t{};
  for (int i = 0; i < E::ND; ++i) {
    double s = 0.0;
    for (int q = 0; q < E::NQ; ++q) {
      t.wphi[i][q] = E::w(q) * E::phi(q, i);
      s += t.wphi[i][q];
      for (int b = 0; b <= GDIM; ++b) t.dphi[i][q][b] = E::dphi(q, i, b);
    }
    t.iphi[i] = s;
  }
  return t;
}
__device__ const RtTab<2, 1> RT21 = make_rt<2, 1>();
__device__ const RtTab<2, 2> RT22 = make_rt<2, 2>();
__device__ const RtTab<3, 1> RT31 = make_rt<3, 1>();
__device__ const RtTab<3, 2> RT32 = make_rt<3, 2>();
__device__ const RtTab<2, 3, 1> RT23 = make_rt<2, 3, 1>();
__device__ const RtTab<2, 2, 1> RT22H = make_rt<2, 2, 1>();  // P2 rows next to a P3 velocity (div(u) q, rows Q)
__device__ const RtTab<3, 3, 1> RT33 = make_rt<3, 3, 1>();   // P3 on tetrahedra (round 5)
__device__ const RtTab<3, 2, 1> RT32H = make_rt<3, 2, 1>();
template <int GDIM, int DEG, int RULE = (DEG == 3 ? 1 : 0)>
__device__ __forceinline__ const RtTab<GDIM, DEG, RULE> &rt() {
  if constexpr (RULE == 1 && DEG == 3 && GDIM == 3) return RT33;
  else if constexpr (RULE == 1 && DEG == 3) return RT23;
  else if constexpr (RULE == 1) {
    static_assert(DEG == 2, "row tables on the high-order rule: P2 and P3");
    if constexpr (GDIM == 3) return RT32H;
    else return RT22H;
  } else if constexpr (GDIM == 2 && DEG == 1) return RT21;
  else if constexpr (GDIM == 2 && DEG == 2) return RT22;
  else if constexpr (GDIM == 3 && DEG == 1) return RT31;
  else return RT32;
}

// Convection as a tensor contraction (assemble_first, fracstep.py:355-358).  With u_ab in the element's
// own space, beta_b(x) = G[b] . u_ab(x) = sum_k phi_k(x) gam[k][b], gam[k][b] = G[b] . u_ab_k, and
//   C[i][j] = |J| sum_b int phi_i beta_b d(phi_j)/d(lambda_b) = |J| sum_{(j,b)} sum_k T[i][(j,b)][k] gam[k][b]
// where T[i][(j,b)][k] = int phi_i phi_k d(phi_j)/d(lambda_b) over the reference simplex -- computed at
// compile time with the same degree-5 rule (exact: the integrand has degree 2+2+1), and only for the
// (j, b) pairs whose derivative is not identically zero (16 of 40 for P2 tetrahedra).  A lane needs
// row i of T for ITS row dof: 160 doubles read from an LDS copy, against 590 FMAs per (row, cell)
// pair that the quadrature form spent on evaluating u_ab at the 14 points again for every row of
// the cell (870 -> 280 FMAs per pair).
template <int GDIM, int DEG>
struct Combos {
  static constexpr int MAXC = Elem<GDIM, DEG>::ND * (GDIM + 1);
  int n;
  int j[MAXC], b[MAXC];
};
template <int GDIM, int DEG>
constexpr Combos<GDIM, DEG> make_combos() {
  using E = Elem<GDIM, DEG>;
  Combos<GDIM, DEG> c{};
  for (int j = 0; j < E::ND; ++j)
    for (int b = 0; b <= GDIM; ++b) {
      bool nz = false;
      for (int q = 0; q < E::NQ; ++q) nz = nz || (E::dphi(q, j, b) != 0.0);
      if (nz) {
        c.j[c.n] = j;
        c.b[c.n] = b;
        ++c.n;
      }
    }
  return c;
}
template <int GDIM, int DEG>
inline constexpr Combos<GDIM, DEG> COMBOS = make_combos<GDIM, DEG>();

template <int GDIM, int DEG>
struct ConvTab {
  using E = Elem<GDIM, DEG>;
  static constexpr int NCB = COMBOS<GDIM, DEG>.n;
  double t[E::ND][E::ND][NCB];  // [row dof i][k][(j, b)]: read in this order, 16 independent chains per k
};
template <int GDIM, int DEG>
constexpr ConvTab<GDIM, DEG> make_conv() {
  using E = Elem<GDIM, DEG>;
  ConvTab<GDIM, DEG> T{};
  constexpr auto CB = COMBOS<GDIM, DEG>;
  for (int i = 0; i < E::ND; ++i)
    for (int m = 0; m < CB.n; ++m)
      for (int k = 0; k < E::ND; ++k) {
        double s = 0.0;
        for (int q = 0; q < E::NQ; ++q) s += E::w(q) * E::phi(q, i) * E::phi(q, k) * E::dphi(q, CB.j[m], CB.b[m]);
        T.t[i][k][m] = s;
      }
  return T;
}
__device__ const ConvTab<2, 1> CT21 = make_conv<2, 1>();
__device__ const ConvTab<2, 2> CT22 = make_conv<2, 2>();
__device__ const ConvTab<3, 1> CT31 = make_conv<3, 1>();
__device__ const ConvTab<3, 2> CT32 = make_conv<3, 2>();
__device__ const ConvTab<2, 3> CT23 = make_conv<2, 3>();
template <int GDIM, int DEG>
__device__ __forceinline__ const ConvTab<GDIM, DEG> &ct() {
  static_assert(!(GDIM == 3 && DEG == 3), "P3 tetrahedra: convection by quadrature (QuadTab33), no tensor table");
  if constexpr (DEG == 3) return CT23;
  else if constexpr (GDIM == 2 && DEG == 1) return CT21;
  else if constexpr (GDIM == 2 && DEG == 2) return CT22;
  else if constexpr (GDIM == 3 && DEG == 1) return CT31;
  else return CT32;
}

// P3 on tetrahedra: the tensor T[i][k][(j, b)] the kernels above read from LDS would be 20 x 20 x 60 doubles = 192 KB --
// it does not fit.  The convection rows of that element are formed by QUADRATURE at run time instead: the values and
// derivatives of all 20 basis functions at the 70 points sit in device memory and are read with wave-uniform indices
// (scalar loads: every lane of a wave is at the same point q and basis function k / j); only w_q phi_i(x_q) depends on
// the lane (its row dof i) and comes from an LDS copy [20][70].
struct QuadTab33 {
  double phi[OX_NQ3H][20];
  double dphi[OX_NQ3H][20][3];  // d / d lambda_1..3 (the lambda_0 column of the P3 tabulation is identically zero)
};
constexpr QuadTab33 make_qt33() {
  QuadTab33 t{};
  for (int q = 0; q < OX_NQ3H; ++q)
    for (int k = 0; k < 20; ++k) {
      t.phi[q][k] = OX_PHI3H_3[q][k];
      for (int b = 0; b < 3; ++b) t.dphi[q][k][b] = OX_DPHI3H_3[q][k][b + 1];
    }
  return t;
}
__device__ const QuadTab33 QT33 = make_qt33();
template <int GDIM, int DEG>
inline constexpr bool OX_CONV_BY_QUADRATURE = (GDIM == 3 && DEG == 3);

template <int GDIM>
__device__ __forceinline__ void load_geom(const double *__restrict__ g, double (&G)[GDIM + 1][GDIM],
                                          double &adet) {
#pragma unroll
  for (int d = 0; d < GDIM; ++d) G[0][d] = 0.0;
#pragma unroll
  for (int a = 1; a <= GDIM; ++a)
#pragma unroll
    for (int d = 0; d < GDIM; ++d) {
      G[a][d] = g[(a - 1) * GDIM + d];
      G[0][d] -= G[a][d];
    }
  adet = g[GDIM * GDIM];
}

// ---------------------------------------------------------------------------------------
// Matrix rows.  One wave per slice (64-thread blocks, wave-private LDS accumulator
// acc[k][lane] for the slice's rows).  KIND = MASS / STIFF write A.vals = acc; KIND = CONV is
// the fused assemble_first (see oasisx_hip.h).
// ---------------------------------------------------------------------------------------
struct FirstArgs {
  const double *Mv, *Kv, *uab, *u1, *b0;
  double *b_first;
  double *a_u1;  // optional: (new A) @ u1 per row, the same sums an SpMV with the assembled A makes
  double idt, nu;
  // value dictionaries of M and K (la.SellMatrix.freeze): 1-byte codes instead of the f64 values
  const uint8_t *Mc, *Kc;
  const double *Md, *Kd;
  int nMd, nKd;
#ifdef OX_DIAG
  int dbg;  // DIAGNOSTIC BUILDS ONLY (-DOX_DIAG, tools/build_diag.sh; never the product .so): OX_AF_DBG bit 0 skips the
            // pair loop, bit 1 the epilogue -- wrong results, for timing the two halves
#endif
};
#ifdef OX_DIAG
#define OX_AF_SKIP_PAIRS(F) ((F).dbg & 1)
#define OX_AF_SKIP_EPILOGUE(F) ((F).dbg & 2)
#else
#define OX_AF_SKIP_PAIRS(F) false
#define OX_AF_SKIP_EPILOGUE(F) false
#endif

// Two ways of handing slices to waves:
//  * width bins (BLK = false): one launch per bin over a LIST of slices of at most `bin_width` entries, 4 waves per block,
//    accumulators [4][bin_width][64];
//  * row blocks (BLK = true, round 5): ONE launch over the slices IN STORAGE ORDER; `slice_list` is then blk_ptr
//    [n_list + 1]: block b owns the consecutive slices [blk_ptr[b], blk_ptr[b+1]) -- at most 8, one per wave, as many
//    as the launch's LDS holds accumulators for (ox_pattern_info.row_blk_*).  The bins tear the slices of one
//    length-sort window (= one compact region of the mesh) apart into up to ten launches, each of which fetches that
//    region's cell records and coefficients again (refined Delaunay mesh, 18.9 M rows: 95.9 GB of HBM traffic per call
//    for ~22 GB of streams, profiles/r04b_delaunay_pmc_hbm.csv); in storage order the ten rows of a cell pass through
//    one L2 within a few blocks of each other.  Same per-slice arithmetic either way: bit-identical results.
// one slice: the (row, cell) pair loop into the wave-private LDS accumulator `acc`, then the epilogue
template <int GDIM, int DEG, int KIND, int PW, bool DICT, int U>
__device__ __forceinline__ void assemble_slice(const ox_cells &cells, const int32_t *__restrict__ cell_dofs, const ox_adj &adj,
                                               const uint8_t *__restrict__ adj_pos, const ox_sell &A, const FirstArgs &F,
                                               const int slice, double *acc, const int lane, const double *tconv,
                                               const double *dM, const double *dK) {
  using E = Elem<GDIM, DEG>;
  constexpr int ND = E::ND, NQ = E::NQ, GS = E::GS;
  constexpr bool QUAD = OX_CONV_BY_QUADRATURE<GDIM, DEG>;
  constexpr int NCB = QUAD ? 1 : COMBOS<GDIM, DEG>.n;
  constexpr int TS = NCB * ND + 2;
  const int64_t base = A.slice_ptr[slice];
  const int width = (int)((A.slice_ptr[slice + 1] - base) >> 6);
  for (int k = 0; k < width; ++k) acc[k * 64 + lane] = 0.0;
  const int64_t abase = adj.adj_ptr[slice];
  const int T = (int)((adj.adj_ptr[slice + 1] - abase) >> 6);
  const auto &R = rt<GDIM, DEG>();
  // U (row, cell) pairs per lane are in flight together, phase by phase: the three dependent memory
  // rounds of a pair (adjacency -> cell dofs + geometry -> coefficient gathers) are paid once per U
  // pairs.  The wide (vertex-row) bins run ONE wave per SIMD (their accumulators fill the LDS), so
  // nothing else hides those ~3 us: with U = 1 the loop sat at 7 us per pair for 0.5 us of arithmetic.
  for (int t0 = 0; t0 < (OX_AF_SKIP_PAIRS(F) ? 0 : T); t0 += U) {
    int e[U], iloc[U];
    bool ok[U];
    uint8_t pos[U][PW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool inb = t0 + u < T;  // wave-uniform
      const int64_t pidx = abase + (int64_t)(inb ? t0 + u : t0) * 64 + lane;
      e[u] = adj.adj_cell[pidx];
      iloc[u] = adj.adj_loc[pidx];
      ok[u] = inb && e[u] >= 0;
      if (!ok[u]) e[u] = 0;  // loads stay unconditional (cell 0), the result is not accumulated
      if constexpr (PW == 32) {
        *reinterpret_cast<uint4 *>(pos[u]) = *reinterpret_cast<const uint4 *>(adj_pos + pidx * 32);
        *reinterpret_cast<uint4 *>(pos[u] + 16) = *reinterpret_cast<const uint4 *>(adj_pos + pidx * 32 + 16);
      } else if constexpr (PW == 16) {
        *reinterpret_cast<uint4 *>(pos[u]) = *reinterpret_cast<const uint4 *>(adj_pos + pidx * 16);
      } else if constexpr (PW == 8) {
        *reinterpret_cast<uint2 *>(pos[u]) = *reinterpret_cast<const uint2 *>(adj_pos + pidx * 8);
      } else {
        *reinterpret_cast<uint32_t *>(pos[u]) = *reinterpret_cast<const uint32_t *>(adj_pos + pidx * 4);
      }
    }
    double G[U][GDIM + 1][GDIM], adet[U];
    int32_t dd[U][KIND == OX_KIND_CONV ? ND : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#ifdef OX_DIAG
      // (OX_AF_DBG bit 2, diagnostic builds only: every pair reads one of 8 cells' geometry -- L1 hits --: what a
      // dictionary of the cell geometry could save at most)
      load_geom<GDIM>(cells.geom + (size_t)((F.dbg & 32) ? 0 : ((F.dbg & 4) ? (e[u] & 7) : e[u])) * GS, G[u], adet[u]);  // (bit 5: ONE cell: scalar loads)
#else
      load_geom<GDIM>(cells.geom + (size_t)e[u] * GS, G[u], adet[u]);
#endif
      if constexpr (KIND == OX_KIND_CONV) {
#ifdef OX_DIAG
        // (OX_AF_DBG bit 4, diagnostic builds only: NO cell-dof loads -- the gathers go to dofs e, e + 1, ...; with bit 2
        // this prices a per-cell pre-pass that hands the pair loop G.u directly: 8 of its ~31 load instructions gone)
        if (F.dbg & 16) {
#pragma unroll
          for (int k = 0; k < ND; ++k) dd[u][k] = e[u] + k;
        } else
#endif
#pragma unroll
        for (int k = 0; k < ND; ++k) dd[u][k] = cell_dofs[(size_t)e[u] * ND + k];
      }
    }
    double uc[U][KIND == OX_KIND_CONV ? ND : 1][GDIM];
    if constexpr (KIND == OX_KIND_CONV) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < ND; ++k)
#pragma unroll
#ifdef OX_DIAG
          // (OX_AF_DBG bit 3, diagnostic builds only: the coefficient gathers read 16 fixed dofs -- L1 hits --: what any
          // staging of uab (an LDS window per slice, a per-cell pre-pass) could save at most)
          for (int d = 0; d < GDIM; ++d) uc[u][k][d] = F.uab[(size_t)((F.dbg & 8) ? (dd[u][k] & 15) : dd[u][k]) * GDIM + d];
#else
          for (int d = 0; d < GDIM; ++d) uc[u][k][d] = F.uab[(size_t)dd[u][k] * GDIM + d];
#endif
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = iloc[u];
      double c[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j) c[j] = 0.0;
      if constexpr (KIND == OX_KIND_MASS) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const double wp = R.wphi[i][q];
#pragma unroll
          for (int j = 0; j < ND; ++j) c[j] = fma(wp, E::phi(q, j), c[j]);
        }
      } else if constexpr (KIND == OX_KIND_STIFF) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          double gi[GDIM];
#pragma unroll
          for (int d = 0; d < GDIM; ++d) {
            gi[d] = 0.0;
#pragma unroll
            for (int b = 0; b <= GDIM; ++b) gi[d] = fma(R.dphi[i][q][b], G[u][b][d], gi[d]);
          }
          // h[b] = w_q * G[b] . grad(phi_i)
          double h[GDIM + 1];
#pragma unroll
          for (int b = 0; b <= GDIM; ++b) {
            h[b] = 0.0;
#pragma unroll
            for (int d = 0; d < GDIM; ++d) h[b] = fma(G[u][b][d], gi[d], h[b]);
            h[b] *= E::w(q);
          }
#pragma unroll
          for (int j = 0; j < ND; ++j)
#pragma unroll
            for (int b = 0; b <= GDIM; ++b)
              if (E::dphi(q, j, b) != 0.0) c[j] = fma(E::dphi(q, j, b), h[b], c[j]);
        }
      } else if constexpr (QUAD) {
        // convection row of a P3 tetrahedron by quadrature (see QuadTab33): at every point u_ab(x_q) = sum_k phi_k uc_k,
        // beta_b = G[b] . u_ab, and C[i][j] += w_q phi_i(x_q) sum_b beta_b d(phi_j)/d(lambda_b)
        const double *__restrict__ wpi = tconv + i * NQ;  // (LDS: w_q phi_i(x_q) of THIS lane's row dof)
        for (int q = 0; q < NQ; ++q) {
          double uq[GDIM];
#pragma unroll
          for (int d = 0; d < GDIM; ++d) uq[d] = 0.0;
#pragma unroll
          for (int k = 0; k < ND; ++k) {
            const double pk = QT33.phi[q][k];
#pragma unroll
            for (int d = 0; d < GDIM; ++d) uq[d] = fma(pk, uc[u][k][d], uq[d]);
          }
          const double wq = wpi[q];
          double sb[GDIM];  // w phi_i beta_b for b = 1..3 (the lambda_0 derivatives of the tabulation are zero)
#pragma unroll
          for (int b = 1; b <= GDIM; ++b) {
            double v = 0.0;
#pragma unroll
            for (int d = 0; d < GDIM; ++d) v = fma(G[u][b][d], uq[d], v);
            sb[b - 1] = wq * v;
          }
#pragma unroll
          for (int j = 0; j < ND; ++j) {
#pragma unroll
            for (int b = 0; b < GDIM; ++b) c[j] = fma(QT33.dphi[q][j][b], sb[b], c[j]);
          }
        }
      } else {
        // convection row: C[i][j] = int (uab . grad phi_j) phi_i   (fracstep.py:355-358), as the tensor
        // contraction described at ConvTab
        constexpr auto CB = COMBOS<GDIM, DEG>;
        const double *__restrict__ ti = tconv + i * TS;
        double cm[NCB];  // one accumulator per (j, b): NCB independent chains, summed per j at the end
#pragma unroll
        for (int m = 0; m < NCB; ++m) cm[m] = 0.0;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
          double gam[GDIM + 1];
#pragma unroll
          for (int b = 0; b <= GDIM; ++b) {
            double v = 0.0;
#pragma unroll
            for (int d = 0; d < GDIM; ++d) v = fma(G[u][b][d], uc[u][k][d], v);
            gam[b] = v;
          }
#pragma unroll
          for (int m = 0; m < NCB; ++m) cm[m] = fma(ti[k * NCB + m], gam[CB.b[m]], cm[m]);
        }
#pragma unroll
        for (int m = 0; m < NCB; ++m) c[CB.j[m]] += cm[m];
      }
      if (ok[u]) {
#pragma unroll
        for (int j = 0; j < ND; ++j)  // the slot is private to this lane: a no-return ds_add_f64 replaces read+add+write
          __hip_atomic_fetch_add(&acc[(int)pos[u][j] * 64 + lane], adet[u] * c[j], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
  }
  // epilogue: stream the slice (coalesced 16-B values, 8-B columns)
  const int64_t row = (int64_t)slice * 64 + lane;
  const int npair = OX_AF_SKIP_EPILOGUE(F) ? 0 : width >> 1;
  double2 *__restrict__ av = reinterpret_cast<double2 *>(A.vals + base) + lane;
  if constexpr (KIND != OX_KIND_CONV) {
    for (int k = 0; k < npair; ++k) {
      double2 v;
      v.x = acc[(2 * k) * 64 + lane];
      v.y = acc[(2 * k + 1) * 64 + lane];
      av[(size_t)k * 64] = v;
    }
  } else {
    const double2 *__restrict__ mv = reinterpret_cast<const double2 *>(F.Mv + base) + lane;
    const double2 *__restrict__ kv = reinterpret_cast<const double2 *>(F.Kv + base) + lane;
    const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
    // (the pattern's 16-bit column stream was tried here -- 2 instead of 4 B per entry of ~14 B of streams --
    // and measured SLOWER: 9.25 vs 8.50 ms per assemble_first on one box, tools/af_bench.py; the decode's scalar
    // base loads and selects cost the epilogue more than the 1 GB it saves)
    const double idt = F.idt, mhnu = -0.5 * F.nu;
    double bf[GDIM], af[GDIM];
#pragma unroll
    for (int d = 0; d < GDIM; ++d) bf[d] = af[d] = 0.0;
    const bool want_au = F.a_u1 != nullptr;
    const unsigned short *__restrict__ mc =
        DICT ? reinterpret_cast<const unsigned short *>(F.Mc + base) + lane : nullptr;
    const unsigned short *__restrict__ kc =
        DICT ? reinterpret_cast<const unsigned short *>(F.Kc + base) + lane : nullptr;
    // 4 entry pairs per turn, phase by phase (codes + columns -> u1 gathers -> arithmetic in the
    // original order -> stores): one turn pays the two dependent memory rounds once instead of 4 times
    constexpr int EU = 4;
    for (int k0 = 0; k0 < npair; k0 += EU) {
      double2 m[EU], kk[EU];
      int2 col[EU];
#pragma unroll
      for (int q = 0; q < EU; ++q) {
        const int k = min(k0 + q, npair - 1);  // unconditional loads; the surplus is not used
        if constexpr (DICT) {
          const unsigned cm = mc[(size_t)k * 64], ck = kc[(size_t)k * 64];
          m[q].x = dM[cm & 0xff], m[q].y = dM[cm >> 8];
          kk[q].x = dK[ck & 0xff], kk[q].y = dK[ck >> 8];
        } else {
          m[q] = mv[(size_t)k * 64], kk[q] = kv[(size_t)k * 64];
        }
        col[q] = cp[(size_t)k * 64];
      }
      double x0[EU][GDIM], x1[EU][GDIM];
#pragma unroll
      for (int q = 0; q < EU; ++q)
#pragma unroll
        for (int d = 0; d < GDIM; ++d) {
          x0[q][d] = F.u1[(size_t)col[q].x * GDIM + d];
          x1[q][d] = F.u1[(size_t)col[q].y * GDIM + d];
        }
#pragma unroll
      for (int q = 0; q < EU; ++q) {
        const int k = k0 + q;
        if (k < npair) {
          const double c0 = acc[(2 * k) * 64 + lane], c1 = acc[(2 * k + 1) * 64 + lane];
          // A = -0.5 C; A += (1/dt) M; A += (-0.5 nu) K        (fracstep.py:438-442)
          const double ar0 = fma(mhnu, kk[q].x, fma(idt, m[q].x, -0.5 * c0));
          const double ar1 = fma(mhnu, kk[q].y, fma(idt, m[q].y, -0.5 * c1));
#pragma unroll
          for (int d = 0; d < GDIM; ++d) bf[d] = fma(ar1, x1[q][d], fma(ar0, x0[q][d], bf[d]));
          // A = -A; A += (2/dt) M                               (fracstep.py:468-469)
          double2 v;
          v.x = fma(2.0 * idt, m[q].x, -ar0);
          v.y = fma(2.0 * idt, m[q].y, -ar1);
          av[(size_t)k * 64] = v;
          if (want_au) {  // entry order and operations of k_spmv: bit-identical to A.mult(u1)
#pragma unroll
            for (int d = 0; d < GDIM; ++d) af[d] = fma(v.y, x1[q][d], fma(v.x, x0[q][d], af[d]));
          }
        }
      }
    }
    if (row < A.n_rows) {
#pragma unroll
      for (int d = 0; d < GDIM; ++d)
        F.b_first[row * GDIM + d] = bf[d] + F.b0[row * GDIM + d];  // + b0 (fracstep.py:456)
      if (want_au) {
#pragma unroll
        for (int d = 0; d < GDIM; ++d) F.a_u1[row * GDIM + d] = af[d];
      }
    }
  }
}

template <int GDIM, int DEG, int KIND, int PW, bool DICT = false, int U = 1, bool BLK = false>
__global__ __launch_bounds__(BLK ? 512 : 256) void k_assemble_rows(ox_cells cells, const int32_t *__restrict__ cell_dofs,
                                                       ox_adj adj, const uint8_t *__restrict__ adj_pos,
                                                       ox_sell A, FirstArgs F,
                                                       const int32_t *__restrict__ slice_list, int n_list,
                                                       int bin_width) {
  using E = Elem<GDIM, DEG>;
  constexpr int ND = E::ND;
  constexpr bool QUAD = OX_CONV_BY_QUADRATURE<GDIM, DEG>;
  constexpr int NCB = QUAD ? 1 : COMBOS<GDIM, DEG>.n;
  constexpr int TS = NCB * ND + 2;  // doubles per row dof in LDS: 16 B of padding put the rows of
                                    // different i on different banks for the 16-byte reads
  extern __shared__ double acc_all[];  // [4 waves][bin_width][64]  (BLK: the row block's slices back to back)
  __shared__ double dM[DICT ? 256 : 1], dK[DICT ? 256 : 1];
  // (P3 tetrahedra: w_q phi_i(x_q), [ND][NQ], for the quadrature form of the convection rows)
  __shared__ __attribute__((aligned(16))) double tconv[KIND == OX_KIND_CONV ? (QUAD ? ND * E::NQ : ND * TS) : 2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nthr = blockDim.x, nwave = blockDim.x >> 6;  // 4 waves per block, fewer for very wide rows
  if constexpr (DICT) {
    for (int i = threadIdx.x; i < F.nMd; i += nthr) dM[i] = F.Md[i];
    for (int i = threadIdx.x; i < F.nKd; i += nthr) dK[i] = F.Kd[i];
  }
  if constexpr (KIND == OX_KIND_CONV && QUAD) {
    const auto &Rq = rt<GDIM, DEG>();
    for (int idx = threadIdx.x; idx < ND * E::NQ; idx += nthr) tconv[idx] = Rq.wphi[idx / E::NQ][idx % E::NQ];
  } else if constexpr (KIND == OX_KIND_CONV) {
    const double *src = &ct<GDIM, DEG>().t[0][0][0];
    constexpr int NT = ND * NCB * ND, PER = (NT + 63) / 64;  // loads per thread of a 64-thread block
    double tv[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {  // all loads first (one memory round trip), then the LDS writes
      const int idx = threadIdx.x + r * nthr;
      tv[r] = idx < NT ? src[idx] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int idx = threadIdx.x + r * nthr;
      if (idx < NT) {
        const int i = idx / (NCB * ND);
        tconv[i * TS + (idx - i * NCB * ND)] = tv[r];
      }
    }
  }
  if constexpr (DICT || KIND == OX_KIND_CONV) __syncthreads();
  if constexpr (BLK) {
    // one block per row block; the blocks of one XCD take a contiguous eighth of them.  (A PERSISTENT grid -- one block per
    // compute unit walking an interleaved share of the row blocks, tables staged once -- was measured and dropped: the
    // static shares are uneven and every row block ends in a barrier: 128^3 box 7.83 -> 7.98 ms, refined Delaunay mesh
    // 13.5 -> 16.9 ms.)
    const int b = ox_xcd_remap(blockIdx.x, gridDim.x);  // (gridDim.x == n_list)
    const int s0 = slice_list[b];
    const int slice = s0 + wave;
    if (slice >= slice_list[b + 1]) return;
    // (consecutive slices: the entries before this one)
    assemble_slice<GDIM, DEG, KIND, PW, DICT, U>(cells, cell_dofs, adj, adj_pos, A, F, slice,
                                                 acc_all + (A.slice_ptr[slice] - A.slice_ptr[s0]), lane, tconv, dM, dK);
  } else {
    // 4 slices of the bin per block; blocks that share an XCD (equal blockIdx % 8) take one contiguous
    // eighth of the bin's slices: the rows of a cell (its 4 vertices / 6 edges) then meet in ONE L2
    // instead of being fetched by up to 8 (r01 PMC: 46.6 GB per assemble_first, every pair re-fetched
    // its cell; with the chunked order 24.4 GB)
    const int li = ox_xcd_remap(blockIdx.x, gridDim.x) * nwave + wave;
    if (li >= n_list) return;
    assemble_slice<GDIM, DEG, KIND, PW, DICT, U>(cells, cell_dofs, adj, adj_pos, A, F, slice_list[li],
                                                 acc_all + (size_t)wave * bin_width * 64, lane, tconv, dM, dK);
  }
}

template <int GDIM, int DEG, int KIND, int PW, bool DICT = false>
static int launch_rows_t(const ox_cells *cells, const int32_t *cell_dofs, const ox_adj *adj,
                         const uint8_t *adj_pos, const ox_sell *A, const FirstArgs &F, int n_bins,
                         const int64_t *bin_ptr, const int32_t *bin_slices, const int32_t *bin_width,
                         hipStream_t st) {
  if constexpr (KIND == OX_KIND_CONV && !DICT) {
    if (F.Mc && F.Kc && F.Md && F.Kd && F.nMd >= 1 && F.nMd <= 256 && F.nKd >= 1 && F.nKd <= 256)
      return launch_rows_t<GDIM, DEG, KIND, PW, true>(cells, cell_dofs, adj, adj_pos, A, F, n_bins, bin_ptr,
                                                      bin_slices, bin_width, st);
  }
  for (int b = 0; b < n_bins; ++b) {
    const int64_t cnt = bin_ptr[b + 1] - bin_ptr[b];
    if (cnt <= 0) continue;
    // wave-private accumulators [width][64]; 4 waves per block unless the rows are so wide (unstructured
    // meshes: > 70 entries) that fewer fit beside the 17 KB of tables
    int nw = 4;
    while (nw > 1 && (size_t)nw * bin_width[b] * 64 * sizeof(double) > 140 * 1024) nw >>= 1;
    const size_t lds = (size_t)nw * bin_width[b] * 64 * sizeof(double);
    if (lds > 140 * 1024) OX_FAIL("assemble: row width %d needs %zu B of LDS", bin_width[b], lds);
    // (U = 2 / 3 (row, cell) pairs in flight per lane were measured at 128^3 -- 9.0 / 9.3 ms against 7.7 with one: the
    // pair loop is bound by the texture path, every lane gathering from another cell, not by latency -- and their
    // instantiations and the OX_ASSEMBLE_U / OX_ASSEMBLE_NW tuning switches removed in round 5)
    auto go = [&](auto kern) -> int {
      if (lds > 32 * 1024)
        OX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3((unsigned)((cnt + nw - 1) / nw)), dim3(64 * nw), lds, st, *cells, cell_dofs, *adj,
                         adj_pos, *A, F, bin_slices + bin_ptr[b], (int)cnt, (int)bin_width[b]);
      OX_LAUNCH_CHECK();
      return 0;
    };
    const int rc = go(k_assemble_rows<GDIM, DEG, KIND, PW, DICT, 1>);
    if (rc) return rc;
  }
  return 0;
}

// row-block launch (k_assemble_rows<..., BLK = true>): one launch, 8 waves per block
template <int GDIM, int DEG, int KIND, int PW, bool DICT = false>
static int launch_row_blocks_t(const ox_cells *cells, const int32_t *cell_dofs, const ox_adj *adj, const uint8_t *adj_pos,
                               const ox_sell *A, const FirstArgs &F, int n_blocks, const int32_t *blk_ptr, int64_t lds_entries,
                               hipStream_t st) {
  if constexpr (KIND == OX_KIND_CONV && !DICT) {
    if (F.Mc && F.Kc && F.Md && F.Kd && F.nMd >= 1 && F.nMd <= 256 && F.nKd >= 1 && F.nKd <= 256)
      return launch_row_blocks_t<GDIM, DEG, KIND, PW, true>(cells, cell_dofs, adj, adj_pos, A, F, n_blocks, blk_ptr, lds_entries, st);
  }
  if (n_blocks <= 0) return 0;
  const size_t lds = (size_t)lds_entries * sizeof(double);
  if (lds > OX_ROW_BLOCK_LDS) OX_FAIL("assemble: a row block needs %zu B of LDS (limit %d)", lds, OX_ROW_BLOCK_LDS);
  auto kern = k_assemble_rows<GDIM, DEG, KIND, PW, DICT, 1, true>;
  if (lds > 32 * 1024)
    OX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks), dim3(512), lds, st, *cells, cell_dofs, *adj, adj_pos, *A, F, blk_ptr,
                     n_blocks, 0);
  OX_LAUNCH_CHECK();
  return 0;
}
template <int KIND>
static int launch_row_blocks(int degree, const ox_cells *cells, const int32_t *cell_dofs, const ox_adj *adj,
                             const uint8_t *adj_pos, int pw, const ox_sell *A, const FirstArgs &F, int n_blocks,
                             const int32_t *blk_ptr, int64_t lds_entries, hipStream_t st) {
  const int g = cells->gdim;
#define OX_ROWS_CASE(GD, DG, P)                                                                                       \
  if (g == GD && degree == DG) {                                                                                      \
    if (pw != P) OX_FAIL("assemble: adj_pos stride %d, expected %d", pw, P);                                          \
    return launch_row_blocks_t<GD, DG, KIND, P>(cells, cell_dofs, adj, adj_pos, A, F, n_blocks, blk_ptr, lds_entries, st); \
  }
  OX_ROWS_CASE(2, 1, 4) OX_ROWS_CASE(2, 2, 8) OX_ROWS_CASE(3, 1, 4) OX_ROWS_CASE(3, 2, 16) OX_ROWS_CASE(2, 3, 16) OX_ROWS_CASE(3, 3, 32)
#undef OX_ROWS_CASE
  OX_FAIL("assemble: unsupported gdim=%d degree=%d", g, degree);
}

template <int KIND>
static int launch_rows(int degree, const ox_cells *cells, const int32_t *cell_dofs, const ox_adj *adj,
                       const uint8_t *adj_pos, int pw, const ox_sell *A, const FirstArgs &F,
                       int n_bins, const int64_t *bin_ptr, const int32_t *bin_slices,
                       const int32_t *bin_width, hipStream_t st) {
  const int g = cells->gdim;
#define OX_ROWS_CASE(GD, DG, P)                                                               \
  if (g == GD && degree == DG) {                                                              \
    if (pw != P) OX_FAIL("assemble: adj_pos stride %d, expected %d", pw, P);                  \
    return launch_rows_t<GD, DG, KIND, P>(cells, cell_dofs, adj, adj_pos, A, F, n_bins, bin_ptr, \
                                          bin_slices, bin_width, st);                         \
  }
  OX_ROWS_CASE(2, 1, 4) OX_ROWS_CASE(2, 2, 8) OX_ROWS_CASE(3, 1, 4) OX_ROWS_CASE(3, 2, 16) OX_ROWS_CASE(2, 3, 16) OX_ROWS_CASE(3, 3, 32)
#undef OX_ROWS_CASE
  OX_FAIL("assemble: unsupported gdim=%d degree=%d", g, degree);
}

extern "C" int ox_assemble_matrix(int kind, int degree, const ox_cells *cells, const int32_t *cell_dofs,
                                  const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                                  int n_bins, const int64_t *bin_ptr_host, const int32_t *bin_slices,
                                  const int32_t *bin_width_host, void *stream) {
  if (!cells || !cell_dofs || !adj || !adj_pos || !A) OX_FAIL("ox_assemble_matrix: null argument");
  FirstArgs F{};
  hipStream_t st = ox_stream(stream);
  if (kind == OX_KIND_MASS)
    return launch_rows<OX_KIND_MASS>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_bins,
                                     bin_ptr_host, bin_slices, bin_width_host, st);
  if (kind == OX_KIND_STIFF)
    return launch_rows<OX_KIND_STIFF>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_bins,
                                      bin_ptr_host, bin_slices, bin_width_host, st);
  OX_FAIL("ox_assemble_matrix: kind=%d", kind);
}

extern "C" int ox_assemble_first_au(int degree, const ox_cells *cells, const int32_t *cell_dofs,
                                    const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                                    const ox_sell *M, const ox_sell *K, const double *uab,
                                    const double *u1, const double *b0, double *b_first, double dt,
                                    double nu, int n_bins, const int64_t *bin_ptr_host,
                                    const int32_t *bin_slices, const int32_t *bin_width_host,
                                    void *stream, double *a_u1) {
  if (!cells || !cell_dofs || !adj || !adj_pos || !A || !M || !K || !M->vals || !K->vals || !uab || !u1 ||
      !b0 || !b_first)
    OX_FAIL("ox_assemble_first: null argument");
  if (M->slice_ptr != A->slice_ptr || K->slice_ptr != A->slice_ptr)
    OX_FAIL("ox_assemble_first: M, K and A must share one sparsity pattern");
  if (!(dt > 0.0)) OX_FAIL("ox_assemble_first: dt=%g", dt);
  FirstArgs F{M->vals, K->vals, uab, u1, b0, b_first, a_u1, 1.0 / dt, nu,
              M->vcode, K->vcode, M->vdict, K->vdict, M->n_dict, K->n_dict};
#ifdef OX_DIAG
  {
    const char *e = getenv("OX_AF_DBG");
    F.dbg = e ? atoi(e) : 0;
  }
#endif
  if (ox_prof_on) ox_prof_start(OX_TAG_ASSEMBLE_FIRST, ox_stream(stream));
  const int rc = launch_rows<OX_KIND_CONV>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_bins,
                                           bin_ptr_host, bin_slices, bin_width_host, ox_stream(stream));
  if (ox_prof_on) ox_prof_stop(ox_stream(stream));
  return rc;
}

extern "C" int ox_assemble_first(int degree, const ox_cells *cells, const int32_t *cell_dofs,
                                 const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                                 const ox_sell *M, const ox_sell *K, const double *uab,
                                 const double *u1, const double *b0, double *b_first, double dt,
                                 double nu, int n_bins, const int64_t *bin_ptr_host,
                                 const int32_t *bin_slices, const int32_t *bin_width_host,
                                 void *stream) {
  return ox_assemble_first_au(degree, cells, cell_dofs, adj, adj_pos, pw, A, M, K, uab, u1, b0, b_first, dt, nu, n_bins,
                              bin_ptr_host, bin_slices, bin_width_host, stream, nullptr);
}

extern "C" int ox_assemble_matrix_blocks(int kind, int degree, const ox_cells *cells, const int32_t *cell_dofs,
                                         const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A, int n_blocks,
                                         const int32_t *blk_ptr, int64_t lds_entries, void *stream) {
  if (!cells || !cell_dofs || !adj || !adj_pos || !A || (n_blocks > 0 && !blk_ptr)) OX_FAIL("ox_assemble_matrix_blocks: null argument");
  FirstArgs F{};
  hipStream_t st = ox_stream(stream);
  if (kind == OX_KIND_MASS)
    return launch_row_blocks<OX_KIND_MASS>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_blocks, blk_ptr, lds_entries, st);
  if (kind == OX_KIND_STIFF)
    return launch_row_blocks<OX_KIND_STIFF>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_blocks, blk_ptr, lds_entries, st);
  OX_FAIL("ox_assemble_matrix_blocks: kind=%d", kind);
}

extern "C" int ox_assemble_first_blocks(int degree, const ox_cells *cells, const int32_t *cell_dofs, const ox_adj *adj,
                                        const uint8_t *adj_pos, int pw, const ox_sell *A, const ox_sell *M, const ox_sell *K,
                                        const double *uab, const double *u1, const double *b0, double *b_first, double dt,
                                        double nu, int n_blocks, const int32_t *blk_ptr, int64_t lds_entries, void *stream,
                                        double *a_u1) {
  if (!cells || !cell_dofs || !adj || !adj_pos || !A || !M || !K || !M->vals || !K->vals || !uab || !u1 || !b0 ||
      !b_first || (n_blocks > 0 && !blk_ptr))
    OX_FAIL("ox_assemble_first_blocks: null argument");
  if (M->slice_ptr != A->slice_ptr || K->slice_ptr != A->slice_ptr)
    OX_FAIL("ox_assemble_first: M, K and A must share one sparsity pattern");
  if (!(dt > 0.0)) OX_FAIL("ox_assemble_first: dt=%g", dt);
  FirstArgs F{M->vals, K->vals, uab, u1, b0, b_first, a_u1, 1.0 / dt, nu,
              M->vcode, K->vcode, M->vdict, K->vdict, M->n_dict, K->n_dict};
#ifdef OX_DIAG
  {
    const char *e = getenv("OX_AF_DBG");
    F.dbg = e ? atoi(e) : 0;
  }
#endif
  if (ox_prof_on) ox_prof_start(OX_TAG_ASSEMBLE_FIRST, ox_stream(stream));
  const int rc = launch_row_blocks<OX_KIND_CONV>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_blocks, blk_ptr, lds_entries,
                                                 ox_stream(stream));
  if (ox_prof_on) ox_prof_stop(ox_stream(stream));
  return rc;
}

// ---------------------------------------------------------------------------------------
// Vectors.  lane = row, 4 slices per 256-thread block, no LDS.
// ---------------------------------------------------------------------------------------
#define OX_ADJ_WALK_BEGIN                                                       \
  const int b_ = ox_xcd_remap(blockIdx.x, gridDim.x);                           \
  const int slice = b_ * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;       \
  if (slice >= adj.n_slices) return;                                            \
  const int64_t row = (int64_t)slice * 64 + lane;                               \
  const int64_t abase = adj.adj_ptr[slice];                                     \
  const int T = (int)((adj.adj_ptr[slice + 1] - abase) >> 6);                   \
  for (int t = 0; t < T; ++t) {                                                 \
    const int64_t pidx = abase + (int64_t)t * 64 + lane;                        \
    const int e = adj.adj_cell[pidx];                                           \
    if (e < 0) continue;                                                        \
    const int i = adj.adj_loc[pidx];
#define OX_ADJ_WALK_END }

template <int GDIM, int DEG>
__global__ __launch_bounds__(256) void k_weights(ox_cells cells, ox_adj adj, int64_t n_rows, double *w) {
  using E = Elem<GDIM, DEG>;
  const auto &R = rt<GDIM, DEG>();
  double s = 0.0;
  OX_ADJ_WALK_BEGIN
  s = fma(cells.geom[(size_t)e * E::GS + GDIM * GDIM], R.iphi[i], s);
  OX_ADJ_WALK_END
  if (row < n_rows) w[row] = s;
}

extern "C" int ox_assemble_weights(int degree, const ox_cells *cells, const ox_adj *adj, int64_t n_rows,
                                   double *w, void *stream) {
  if (!cells || !adj || !w) OX_FAIL("ox_assemble_weights: null argument");
  const int nblk = (adj->n_slices + 3) / 4;
  if (nblk == 0) return 0;
  hipStream_t st = ox_stream(stream);
  const int g = cells->gdim;
#define OX_W_CASE(GD, DG)                                                                          \
  if (g == GD && degree == DG) {                                                                   \
    hipLaunchKernelGGL((k_weights<GD, DG>), dim3(nblk), dim3(256), 0, st, *cells, *adj, n_rows, w); \
    OX_LAUNCH_CHECK();                                                                             \
    return 0;                                                                                      \
  }
  OX_W_CASE(2, 1) OX_W_CASE(2, 2) OX_W_CASE(3, 1) OX_W_CASE(3, 2) OX_W_CASE(2, 3) OX_W_CASE(3, 3)
#undef OX_W_CASE
  OX_FAIL("ox_assemble_weights: unsupported gdim=%d degree=%d", g, degree);
}

// b[row] = sum_(cell e, local i) |det J_e| sum_q wphi[q][i] fq[e][q]: the load vector of a tabulated source
// (include/oasisx_hip.h).  The reference table sits in LDS; every lane walks its own row's cells.
template <bool STAGE>
__global__ __launch_bounds__(256) void k_load_vector(ox_cells cells, ox_adj adj, int64_t n_rows, int n_d, int n_q,
                                                     const double *__restrict__ wphi, const double *__restrict__ fq,
                                                     double *__restrict__ b) {
  extern __shared__ double tab_lds[];  // [n_q][n_d] (STAGE); a table beyond 48 KB is read in place (set-up only)
  const double *__restrict__ tab = STAGE ? tab_lds : wphi;
  if (STAGE) {
    for (int k = threadIdx.x; k < n_q * n_d; k += blockDim.x) tab_lds[k] = wphi[k];
    __syncthreads();
  }
  const int gs = cells.gdim == 2 ? 6 : 10, ia = cells.gdim * cells.gdim;
  double s = 0.0;
  OX_ADJ_WALK_BEGIN
  const double *__restrict__ f = fq + (size_t)e * n_q;
  double v = 0.0;
  for (int q = 0; q < n_q; ++q) v = fma(tab[q * n_d + i], f[q], v);
  s = fma(cells.geom[(size_t)e * gs + ia], v, s);
  OX_ADJ_WALK_END
  if (row < n_rows) b[row] = s;
}

extern "C" int ox_assemble_load_vector(const ox_cells *cells, const ox_adj *adj, int64_t n_rows, int n_d, int n_q,
                                       const double *wphi, const double *fq, double *b, void *stream) {
  if (!cells || !adj || !wphi || !fq || !b) OX_FAIL("ox_assemble_load_vector: null argument");
  if (cells->gdim != 2 && cells->gdim != 3) OX_FAIL("ox_assemble_load_vector: gdim=%d", cells->gdim);
  if (n_d < 1 || n_q < 1 || (int64_t)n_d * n_q > (1 << 20)) OX_FAIL("ox_assemble_load_vector: n_d=%d n_q=%d", n_d, n_q);
  const int nblk = (adj->n_slices + 3) / 4;
  if (nblk == 0) return 0;
  if ((int64_t)n_d * n_q <= 6144)
    hipLaunchKernelGGL(k_load_vector<true>, dim3(nblk), dim3(256), (size_t)n_d * n_q * sizeof(double), ox_stream(stream), *cells,
                       *adj, n_rows, n_d, n_q, wphi, fq, b);
  else
    hipLaunchKernelGGL(k_load_vector<false>, dim3(nblk), dim3(256), 0, ox_stream(stream), *cells, *adj, n_rows, n_d, n_q, wphi,
                       fq, b);
  OX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Projection into the discontinuous P1 space (reference function.py:13-143 with a "DG" 1 target, as its
// test_projector.py builds: basix.ufl.element("DG", cell, 1, shape=(gdim,))).  Dof (cell e, vertex a,
// component c) at (e * (gdim+1) + a) * ncomp + c.  No coupling between cells: the mass matrix is block
// diagonal with the blocks |K| (I + 11^T) / ((d+1)(d+2)), whose inverse is closed form -- the "LU" the
// reference asks PETSc for.  One thread per cell.
//   k_dg1_grad_rhs   b[(e,a)][c] = int_e d_c(u) lambda_a   (the form inner(grad(u), v) * dx; u Lagrange P1 / P2)
//   k_dg1_mass       x = M^-1 b   (INV)   or   b = M x
// ---------------------------------------------------------------------------------------
template <int GDIM, int DEG>
__global__ __launch_bounds__(256) void k_dg1_grad_rhs(ox_cells cells, const int32_t *__restrict__ cell_dofs,
                                                      const double *__restrict__ u, double *__restrict__ b) {
  using E = Elem<GDIM, DEG>;
  using E1 = Elem<GDIM, 1>;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells.n_cells) return;
  double G[GDIM + 1][GDIM], adet;
  load_geom<GDIM>(cells.geom + (size_t)e * E::GS, G, adet);
  double uc[E::ND];
#pragma unroll
  for (int k = 0; k < E::ND; ++k) uc[k] = u[cell_dofs[(size_t)e * E::ND + k]];
  double acc[GDIM + 1][GDIM];
#pragma unroll
  for (int a = 0; a <= GDIM; ++a)
#pragma unroll
    for (int d = 0; d < GDIM; ++d) acc[a][d] = 0.0;
#pragma unroll
  for (int q = 0; q < E::NQ; ++q) {
    double gl[GDIM + 1];  // du/d(lambda_b) at the point
#pragma unroll
    for (int bb = 0; bb <= GDIM; ++bb) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < E::ND; ++k) v = fma(E::dphi(q, k, bb), uc[k], v);
      gl[bb] = v;
    }
#pragma unroll
    for (int d = 0; d < GDIM; ++d) {
      double g = 0.0;
#pragma unroll
      for (int bb = 0; bb <= GDIM; ++bb) g = fma(gl[bb], G[bb][d], g);
#pragma unroll
      for (int a = 0; a <= GDIM; ++a) acc[a][d] = fma(E::w(q) * E1::phi(q, a), g, acc[a][d]);
    }
  }
#pragma unroll
  for (int a = 0; a <= GDIM; ++a)
#pragma unroll
    for (int d = 0; d < GDIM; ++d) b[((size_t)e * (GDIM + 1) + a) * GDIM + d] = adet * acc[a][d];
}

template <int GDIM, bool INV>
__global__ __launch_bounds__(256) void k_dg1_mass(ox_cells cells, int ncomp, const double *__restrict__ in,
                                                  double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells.n_cells) return;
  constexpr int GS = GDIM == 2 ? 6 : 10, NV = GDIM + 1;
  constexpr double fact = GDIM == 2 ? 2.0 : 6.0;
  const double alpha = cells.geom[(size_t)e * GS + GDIM * GDIM] / fact / (double)((GDIM + 1) * (GDIM + 2));  // |K| / ((d+1)(d+2))
  for (int c = 0; c < ncomp; ++c) {
    double v[NV], s = 0.0;
#pragma unroll
    for (int a = 0; a < NV; ++a) {
      v[a] = in[((size_t)e * NV + a) * ncomp + c];
      s += v[a];
    }
#pragma unroll
    for (int a = 0; a < NV; ++a)
      out[((size_t)e * NV + a) * ncomp + c] = INV ? (v[a] - s / (double)(GDIM + 2)) / alpha : alpha * (v[a] + s);
  }
}

extern "C" int ox_dg1_grad_rhs(int u_degree, const ox_cells *cells, const int32_t *cell_dofs, const double *u, double *b,
                               void *stream) {
  if (!cells || !cell_dofs || !u || !b) OX_FAIL("ox_dg1_grad_rhs: null argument");
  if (cells->n_cells <= 0) return 0;
  const dim3 grid((unsigned)((cells->n_cells + 255) / 256));
  hipStream_t st = ox_stream(stream);
  const int g = cells->gdim;
#define OX_DG_CASE(GD, DG)                                                                           \
  if (g == GD && u_degree == DG) {                                                                   \
    hipLaunchKernelGGL((k_dg1_grad_rhs<GD, DG>), grid, dim3(256), 0, st, *cells, cell_dofs, u, b);    \
    OX_LAUNCH_CHECK();                                                                               \
    return 0;                                                                                        \
  }
  OX_DG_CASE(2, 1) OX_DG_CASE(2, 2) OX_DG_CASE(3, 1) OX_DG_CASE(3, 2)
#undef OX_DG_CASE
  OX_FAIL("ox_dg1_grad_rhs: unsupported gdim=%d degree=%d", g, u_degree);
}

extern "C" int ox_dg1_mass(int inverse, const ox_cells *cells, int ncomp, const double *in, double *out, void *stream) {
  if (!cells || !in || !out) OX_FAIL("ox_dg1_mass: null argument");
  if (ncomp < 1) OX_FAIL("ox_dg1_mass: ncomp=%d", ncomp);
  if (cells->n_cells <= 0) return 0;
  const dim3 grid((unsigned)((cells->n_cells + 255) / 256));
  hipStream_t st = ox_stream(stream);
  if (cells->gdim == 2) {
    if (inverse) hipLaunchKernelGGL((k_dg1_mass<2, true>), grid, dim3(256), 0, st, *cells, ncomp, in, out);
    else hipLaunchKernelGGL((k_dg1_mass<2, false>), grid, dim3(256), 0, st, *cells, ncomp, in, out);
  } else if (cells->gdim == 3) {
    if (inverse) hipLaunchKernelGGL((k_dg1_mass<3, true>), grid, dim3(256), 0, st, *cells, ncomp, in, out);
    else hipLaunchKernelGGL((k_dg1_mass<3, false>), grid, dim3(256), 0, st, *cells, ncomp, in, out);
  } else {
    OX_FAIL("ox_dg1_mass: gdim=%d", cells->gdim);
  }
  OX_LAUNCH_CHECK();
  return 0;
}

// kind 0: out[r][d] = base[r][d] + scale * int p * d_d(phi_r)       (A6, fracstep.py:487-497)
// kind 1: out[r][d] = base[r][d] + scale * int d_d(p) * phi_r       (A8, fracstep.py:618)
template <int GDIM, int RDEG, int PDEG, int KIND>
__global__ __launch_bounds__(256) void k_grad_vector(ox_cells cells, const int32_t *__restrict__ cell_pdofs,
                                                     ox_adj adj, int64_t n_rows,
                                                     const double *__restrict__ p,
                                                     const double *__restrict__ base_v, double scale,
                                                     double *__restrict__ out) {
  constexpr int RULE = OX_RULE_OF(RDEG, PDEG);
  using ER = Elem<GDIM, RDEG, RULE>;
  using EP = Elem<GDIM, PDEG, RULE>;
  const auto &R = rt<GDIM, RDEG, RULE>();
  double o[GDIM];
#pragma unroll
  for (int d = 0; d < GDIM; ++d) o[d] = 0.0;
  OX_ADJ_WALK_BEGIN
  double G[GDIM + 1][GDIM], adet;
  load_geom<GDIM>(cells.geom + (size_t)e * ER::GS, G, adet);
  const int32_t *__restrict__ pd = cell_pdofs + (size_t)e * EP::ND;
  double pc[EP::ND];
#pragma unroll
  for (int c = 0; c < EP::ND; ++c) pc[c] = p[pd[c]];
  double S[GDIM + 1];
#pragma unroll
  for (int b = 0; b <= GDIM; ++b) S[b] = 0.0;
#pragma unroll
  for (int q = 0; q < ER::NQ; ++q) {
    if constexpr (KIND == 0) {
      double pq = 0.0;
#pragma unroll
      for (int c = 0; c < EP::ND; ++c)
        if (EP::phi(q, c) != 0.0) pq = fma(EP::phi(q, c), pc[c], pq);
      pq *= ER::w(q);
#pragma unroll
      for (int b = 0; b <= GDIM; ++b) S[b] = fma(pq, R.dphi[i][q][b], S[b]);
    } else {
      const double wp = R.wphi[i][q];
#pragma unroll
      for (int b = 0; b <= GDIM; ++b) {
        double gp = 0.0;
#pragma unroll
        for (int c = 0; c < EP::ND; ++c)
          if (EP::dphi(q, c, b) != 0.0) gp = fma(EP::dphi(q, c, b), pc[c], gp);
        S[b] = fma(wp, gp, S[b]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < GDIM; ++d) {
    double v = 0.0;
#pragma unroll
    for (int b = 0; b <= GDIM; ++b) v = fma(S[b], G[b][d], v);
    o[d] = fma(adet, v, o[d]);
  }
  OX_ADJ_WALK_END
  if (row < n_rows) {
#pragma unroll
    for (int d = 0; d < GDIM; ++d)
      out[row * GDIM + d] = fma(scale, o[d], base_v ? base_v[row * GDIM + d] : 0.0);
  }
}

extern "C" int ox_assemble_grad_vector(int kind, int row_degree, int p_degree, const ox_cells *cells,
                                       const int32_t *cell_pdofs, const ox_adj *adj, int64_t n_rows,
                                       const double *p, const double *base, double scale, double *out,
                                       void *stream) {
  if (!cells || !cell_pdofs || !adj || !p || !out) OX_FAIL("ox_assemble_grad_vector: null argument");
  const int nblk = (adj->n_slices + 3) / 4;
  if (nblk == 0) return 0;
  hipStream_t st = ox_stream(stream);
  const int g = cells->gdim;
#define OX_GV_CASE(GD, RD, PD, KD)                                                             \
  if (g == GD && row_degree == RD && p_degree == PD && kind == KD) {                           \
    if (ox_prof_on) ox_prof_start(OX_TAG_GRAD_VECTOR + KD, st);                                \
    hipLaunchKernelGGL((k_grad_vector<GD, RD, PD, KD>), dim3(nblk), dim3(256), 0, st, *cells,  \
                       cell_pdofs, *adj, n_rows, p, base, scale, out);                         \
    if (ox_prof_on) ox_prof_stop(st);                                                          \
    OX_LAUNCH_CHECK();                                                                         \
    return 0;                                                                                  \
  }
#define OX_GV_ALL(GD) \
  OX_GV_CASE(GD, 1, 1, 0) OX_GV_CASE(GD, 1, 1, 1) OX_GV_CASE(GD, 2, 1, 0) OX_GV_CASE(GD, 2, 1, 1)
  OX_GV_ALL(2) OX_GV_ALL(3) OX_GV_CASE(2, 3, 2, 0) OX_GV_CASE(2, 3, 2, 1) OX_GV_CASE(3, 3, 2, 0) OX_GV_CASE(3, 3, 2, 1)
#undef OX_GV_ALL
#undef OX_GV_CASE
  OX_FAIL("ox_assemble_grad_vector: unsupported gdim=%d row_degree=%d p_degree=%d kind=%d", g,
          row_degree, p_degree, kind);
}

// out[r] = scale * int div(u) psi_r                              (A7, fracstep.py:538,546)
template <int GDIM, int RDEG, int UDEG>
__global__ __launch_bounds__(256) void k_div_vector(ox_cells cells, const int32_t *__restrict__ cell_udofs,
                                                    ox_adj adj, int64_t n_rows,
                                                    const double *__restrict__ u, double scale,
                                                    double *__restrict__ out) {
  constexpr int RULE = OX_RULE_OF(RDEG, UDEG);
  using ER = Elem<GDIM, RDEG, RULE>;
  using EU = Elem<GDIM, UDEG, RULE>;
  const auto &R = rt<GDIM, RDEG, RULE>();
  double o = 0.0;
  OX_ADJ_WALK_BEGIN
  double G[GDIM + 1][GDIM], adet;
  load_geom<GDIM>(cells.geom + (size_t)e * ER::GS, G, adet);
  const int32_t *__restrict__ ud = cell_udofs + (size_t)e * EU::ND;
  // gu[k][b] = G[b] . u_k
  double gu[EU::ND][GDIM + 1];
#pragma unroll
  for (int k = 0; k < EU::ND; ++k) {
    const double *up = u + (size_t)ud[k] * GDIM;
    double uk[GDIM];
#pragma unroll
    for (int d = 0; d < GDIM; ++d) uk[d] = up[d];
#pragma unroll
    for (int b = 0; b <= GDIM; ++b) {
      double v = 0.0;
#pragma unroll
      for (int d = 0; d < GDIM; ++d) v = fma(G[b][d], uk[d], v);
      gu[k][b] = v;
    }
  }
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < ER::NQ; ++q) {
    double dv = 0.0;
#pragma unroll
    for (int k = 0; k < EU::ND; ++k)
#pragma unroll
      for (int b = 0; b <= GDIM; ++b)
        if (EU::dphi(q, k, b) != 0.0) dv = fma(EU::dphi(q, k, b), gu[k][b], dv);
    s = fma(R.wphi[i][q], dv, s);
  }
  o = fma(adet, s, o);
  OX_ADJ_WALK_END
  if (row < n_rows) out[row] = scale * o;
}

extern "C" int ox_assemble_div_vector(int row_degree, int u_degree, const ox_cells *cells,
                                      const int32_t *cell_udofs, const ox_adj *adj, int64_t n_rows,
                                      const double *u, double scale, double *out, void *stream) {
  if (!cells || !cell_udofs || !adj || !u || !out) OX_FAIL("ox_assemble_div_vector: null argument");
  const int nblk = (adj->n_slices + 3) / 4;
  if (nblk == 0) return 0;
  hipStream_t st = ox_stream(stream);
  const int g = cells->gdim;
#define OX_DV_CASE(GD, RD, UD)                                                                     \
  if (g == GD && row_degree == RD && u_degree == UD) {                                             \
    if (ox_prof_on) ox_prof_start(OX_TAG_DIV_VECTOR, st);                                          \
    hipLaunchKernelGGL((k_div_vector<GD, RD, UD>), dim3(nblk), dim3(256), 0, st, *cells, cell_udofs, \
                       *adj, n_rows, u, scale, out);                                               \
    if (ox_prof_on) ox_prof_stop(st);                                                              \
    OX_LAUNCH_CHECK();                                                                             \
    return 0;                                                                                      \
  }
  OX_DV_CASE(2, 1, 1) OX_DV_CASE(2, 1, 2) OX_DV_CASE(3, 1, 1) OX_DV_CASE(3, 1, 2) OX_DV_CASE(2, 2, 3) OX_DV_CASE(3, 2, 3)
#undef OX_DV_CASE
  OX_FAIL("ox_assemble_div_vector: unsupported gdim=%d row_degree=%d u_degree=%d", g, row_degree,
          u_degree);
}

// ---------------------------------------------------------------------------------------
// One-off assembly of the rectangular operators (reference fracstep.py:392-404), values stored
// [slot][GDIM].  lane = row; the lane adds every element-matrix row it owns straight into its
// SELL row (the caller passes zero-initialised values) -- plain read-modify-write, single owner.
//   FAM 0 (rows V, cols Q): P[r][c][d] = int psi_c d_d(phi_r)        (p * v.dx(i) * dx, :311-315)
//   FAM 1 (rows V, cols Q): G[r][c][d] = int d_d(psi_c) phi_r        (p.dx(i) * v * dx, :348-352)
//   FAM 2 (rows Q, cols V): D[r][c][d] = int d_d(phi_c) psi_r        (u.dx(i) * q * dx, :332-336)
// ---------------------------------------------------------------------------------------
template <int GDIM, int RDEG, int CDEG, int FAM, int PW>
__global__ __launch_bounds__(256) void k_assemble_rect(ox_cells cells, ox_adj adj,
                                                       const uint8_t *__restrict__ adj_pos, ox_sell A) {
  constexpr int RULE = OX_RULE_OF(RDEG, CDEG);
  using ER = Elem<GDIM, RDEG, RULE>;
  using EC = Elem<GDIM, CDEG, RULE>;
  const auto &R = rt<GDIM, RDEG, RULE>();
  // blocks that share an XCD take one contiguous eighth of the slices (as k_assemble_rows): neighbouring rows,
  // which share their cells, then meet in ONE L2 (r02 PMC: 31-52 GB moved to write 3.3 GB with blocks dealt
  // round-robin over the XCDs)
  const int slice = ox_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= adj.n_slices) return;
  const int64_t base = A.slice_ptr[slice];
  const int64_t abase = adj.adj_ptr[slice];
  const int T = (int)((adj.adj_ptr[slice + 1] - abase) >> 6);
  for (int t = 0; t < T; ++t) {
    const int64_t pidx = abase + (int64_t)t * 64 + lane;
    const int e = adj.adj_cell[pidx];
    if (e < 0) continue;
    const int i = adj.adj_loc[pidx];
    double G[GDIM + 1][GDIM], adet;
    load_geom<GDIM>(cells.geom + (size_t)e * ER::GS, G, adet);
    double c[EC::ND][GDIM];
#pragma unroll
    for (int j = 0; j < EC::ND; ++j)
#pragma unroll
      for (int d = 0; d < GDIM; ++d) c[j][d] = 0.0;
#pragma unroll
    for (int q = 0; q < ER::NQ; ++q) {
      if constexpr (FAM == 0) {  // w psi_j(q) * grad phi_i(q)
        double gi[GDIM];
#pragma unroll
        for (int d = 0; d < GDIM; ++d) {
          gi[d] = 0.0;
#pragma unroll
          for (int b = 0; b <= GDIM; ++b) gi[d] = fma(R.dphi[i][q][b], G[b][d], gi[d]);
          gi[d] *= ER::w(q);
        }
#pragma unroll
        for (int j = 0; j < EC::ND; ++j)
          if (EC::phi(q, j) != 0.0) {
#pragma unroll
            for (int d = 0; d < GDIM; ++d) c[j][d] = fma(EC::phi(q, j), gi[d], c[j][d]);
          }
      } else {  // FAM 1, 2: w phi_i(q) * grad (col basis)_j(q)
        const double wp = R.wphi[i][q];
#pragma unroll
        for (int j = 0; j < EC::ND; ++j)
#pragma unroll
          for (int b = 0; b <= GDIM; ++b)
            if (EC::dphi(q, j, b) != 0.0) {
              const double f = wp * EC::dphi(q, j, b);
#pragma unroll
              for (int d = 0; d < GDIM; ++d) c[j][d] = fma(f, G[b][d], c[j][d]);
            }
      }
    }
#pragma unroll
    for (int j = 0; j < EC::ND; ++j) {
      const int kk = adj_pos[pidx * PW + j];
      double *v = A.vals + ((size_t)base + (size_t)(kk >> 1) * 128 + lane * 2 + (kk & 1)) * GDIM;
#pragma unroll
      for (int d = 0; d < GDIM; ++d) v[d] += adet * c[j][d];
    }
  }
}

extern "C" int ox_assemble_rect(int family, int row_degree, int col_degree, const ox_cells *cells,
                                const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                                void *stream) {
  if (!cells || !adj || !adj_pos || !A) OX_FAIL("ox_assemble_rect: null argument");
  const int nblk = (adj->n_slices + 3) / 4;
  if (nblk == 0) return 0;
  hipStream_t st = ox_stream(stream);
  const int g = cells->gdim;
#define OX_RECT_CASE(GD, RD, CD, FM, P)                                                              \
  if (g == GD && row_degree == RD && col_degree == CD && family == FM) {                             \
    if (pw != P) OX_FAIL("ox_assemble_rect: adj_pos stride %d, expected %d", pw, P);                 \
    hipLaunchKernelGGL((k_assemble_rect<GD, RD, CD, FM, P>), dim3(nblk), dim3(256), 0, st, *cells, *adj, \
                       adj_pos, *A);                                                                 \
    OX_LAUNCH_CHECK();                                                                               \
    return 0;                                                                                        \
  }
#define OX_RECT_DIM(GD, PV2)                                                      \
  OX_RECT_CASE(GD, 1, 1, 0, 4) OX_RECT_CASE(GD, 1, 1, 1, 4) OX_RECT_CASE(GD, 1, 1, 2, 4) \
  OX_RECT_CASE(GD, 2, 1, 0, 4) OX_RECT_CASE(GD, 2, 1, 1, 4) OX_RECT_CASE(GD, 1, 2, 2, PV2)
  OX_RECT_DIM(2, 8) OX_RECT_DIM(3, 16)
  OX_RECT_CASE(2, 3, 2, 0, 8) OX_RECT_CASE(2, 3, 2, 1, 8) OX_RECT_CASE(2, 2, 3, 2, 16)  // P3-P2 on triangles
  OX_RECT_CASE(3, 3, 2, 0, 16) OX_RECT_CASE(3, 3, 2, 1, 16) OX_RECT_CASE(3, 2, 3, 2, 32)  // P3-P2 on tetrahedra
#undef OX_RECT_DIM
#undef OX_RECT_CASE
  OX_FAIL("ox_assemble_rect: unsupported gdim=%d row_degree=%d col_degree=%d family=%d", g, row_degree,
          col_degree, family);
}
