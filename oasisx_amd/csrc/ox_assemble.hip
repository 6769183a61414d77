// Element kernels of the IPCS path, row-centric ("owner computes"): one lane owns one matrix
// row / vector entry and walks the cells adjacent to its dof, so nothing is scattered with
// atomics, every sum has a fixed order (bit-reproducible) and all matrix traffic is
// coalesced in the SELL-64 layout.  Quadrature tables are compile-time constants (fe_tables.h,
// degree-5 rules: every integrand of the path is a polynomial of degree <= 5 on affine cells,
// so the rule is exact, as FFCx's is in the reference).
#include "fe_tables.h"
#include "ox_kernels.h"

#define OX_KIND_MASS 0
#define OX_KIND_STIFF 1
#define OX_KIND_CONV 2

template <int GDIM, int DEG>
struct Elem {
  static constexpr int NV = GDIM + 1;
  static constexpr int ND = DEG == 1 ? GDIM + 1 : (GDIM == 2 ? 6 : 10);
  static constexpr int NQ = GDIM == 2 ? OX_NQ2 : OX_NQ3;
  static constexpr int GS = GDIM == 2 ? 6 : 10;
  __host__ __device__ static constexpr double w(int q) { return GDIM == 2 ? OX_QW2[q] : OX_QW3[q]; }
  __host__ __device__ static constexpr double phi(int q, int k) {
    if constexpr (GDIM == 2 && DEG == 1) return OX_PHI2_1[q][k];
    else if constexpr (GDIM == 2 && DEG == 2) return OX_PHI2_2[q][k];
    else if constexpr (GDIM == 3 && DEG == 1) return OX_PHI3_1[q][k];
    else return OX_PHI3_2[q][k];
  }
  __host__ __device__ static constexpr double dphi(int q, int k, int b) {
    if constexpr (GDIM == 2 && DEG == 1) return OX_DPHI2_1[q][k][b];
    else if constexpr (GDIM == 2 && DEG == 2) return OX_DPHI2_2[q][k][b];
    else if constexpr (GDIM == 3 && DEG == 1) return OX_DPHI3_1[q][k][b];
    else return OX_DPHI3_2[q][k][b];
  }
};

// Tables indexed by a RUNTIME local dof index (the lane's own row dof): device-resident
// copies, w_q*phi_i(q) and dphi_i(q,b).
template <int GDIM, int DEG>
struct RtTab {
  using E = Elem<GDIM, DEG>;
  double wphi[E::ND][E::NQ];
  double dphi[E::ND][E::NQ][GDIM + 1];
  double iphi[E::ND];  // int_ref phi_i
};
template <int GDIM, int DEG>
constexpr RtTab<GDIM, DEG> make_rt() {
  using E = Elem<GDIM, DEG>;
  RtTab<GDIM, DEG> t{};
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
template <int GDIM, int DEG>
__device__ __forceinline__ const RtTab<GDIM, DEG> &rt() {
  if constexpr (GDIM == 2 && DEG == 1) return RT21;
  else if constexpr (GDIM == 2 && DEG == 2) return RT22;
  else if constexpr (GDIM == 3 && DEG == 1) return RT31;
  else return RT32;
}

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
  double idt, nu;
  // value dictionaries of M and K (la.SellMatrix.freeze): 1-byte codes instead of the f64 values
  const uint8_t *Mc, *Kc;
  const double *Md, *Kd;
  int nMd, nKd;
};

template <int GDIM, int DEG, int KIND, int PW, bool DICT = false>
__global__ __launch_bounds__(64) void k_assemble_rows(ox_cells cells, const int32_t *__restrict__ cell_dofs,
                                                      ox_adj adj, const uint8_t *__restrict__ adj_pos,
                                                      ox_sell A, FirstArgs F,
                                                      const int32_t *__restrict__ slice_list) {
  using E = Elem<GDIM, DEG>;
  constexpr int ND = E::ND, NQ = E::NQ, GS = E::GS;
  extern __shared__ double acc[];  // [width][64]
  __shared__ double dM[DICT ? 256 : 1], dK[DICT ? 256 : 1];
  const int lane = threadIdx.x;
  if constexpr (DICT) {
    for (int i = lane; i < F.nMd; i += 64) dM[i] = F.Md[i];
    for (int i = lane; i < F.nKd; i += 64) dK[i] = F.Kd[i];
    __syncthreads();
  }
  const int slice = slice_list[blockIdx.x];
  const int64_t base = A.slice_ptr[slice];
  const int width = (int)((A.slice_ptr[slice + 1] - base) >> 6);
  for (int k = 0; k < width; ++k) acc[k * 64 + lane] = 0.0;
  const int64_t abase = adj.adj_ptr[slice];
  const int T = (int)((adj.adj_ptr[slice + 1] - abase) >> 6);
  const auto &R = rt<GDIM, DEG>();
  for (int t = 0; t < T; ++t) {
    const int64_t pidx = abase + (int64_t)t * 64 + lane;
    const int e = adj.adj_cell[pidx];
    if (e < 0) continue;
    const int i = adj.adj_loc[pidx];
    uint8_t pos[PW];
    if constexpr (PW == 16) {
      *reinterpret_cast<uint4 *>(pos) = *reinterpret_cast<const uint4 *>(adj_pos + pidx * 16);
    } else if constexpr (PW == 8) {
      *reinterpret_cast<uint2 *>(pos) = *reinterpret_cast<const uint2 *>(adj_pos + pidx * 8);
    } else {
      *reinterpret_cast<uint32_t *>(pos) = *reinterpret_cast<const uint32_t *>(adj_pos + pidx * 4);
    }
    double G[GDIM + 1][GDIM], adet;
    load_geom<GDIM>(cells.geom + (size_t)e * GS, G, adet);
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
          for (int b = 0; b <= GDIM; ++b) gi[d] = fma(R.dphi[i][q][b], G[b][d], gi[d]);
        }
        // h[b] = w_q * G[b] . grad(phi_i)
        double h[GDIM + 1];
#pragma unroll
        for (int b = 0; b <= GDIM; ++b) {
          h[b] = 0.0;
#pragma unroll
          for (int d = 0; d < GDIM; ++d) h[b] = fma(G[b][d], gi[d], h[b]);
          h[b] *= E::w(q);
        }
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int b = 0; b <= GDIM; ++b)
            if (E::dphi(q, j, b) != 0.0) c[j] = fma(E::dphi(q, j, b), h[b], c[j]);
      }
    } else {
      // convection row: C[i][j] = int (uab . grad phi_j) phi_i   (fracstep.py:355-358)
      const int32_t *__restrict__ dd = cell_dofs + (size_t)e * ND;
      double uc[ND][GDIM];
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        const double *up = F.uab + (size_t)dd[k] * GDIM;
#pragma unroll
        for (int d = 0; d < GDIM; ++d) uc[k][d] = up[d];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        double ub[GDIM];
#pragma unroll
        for (int d = 0; d < GDIM; ++d) {
          ub[d] = 0.0;
#pragma unroll
          for (int k = 0; k < ND; ++k)
            if (E::phi(q, k) != 0.0) ub[d] = fma(E::phi(q, k), uc[k][d], ub[d]);
        }
        const double wp = R.wphi[i][q];
        double beta[GDIM + 1];
#pragma unroll
        for (int b = 0; b <= GDIM; ++b) {
          beta[b] = 0.0;
#pragma unroll
          for (int d = 0; d < GDIM; ++d) beta[b] = fma(G[b][d], ub[d], beta[b]);
          beta[b] *= wp;
        }
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
          for (int b = 0; b <= GDIM; ++b)
            if (E::dphi(q, j, b) != 0.0) c[j] = fma(E::dphi(q, j, b), beta[b], c[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)  // the slot is private to this lane: a no-return ds_add_f64 replaces read+add+write
      __hip_atomic_fetch_add(&acc[(int)pos[j] * 64 + lane], adet * c[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  // epilogue: stream the slice (coalesced 16-B values, 8-B columns)
  const int64_t row = (int64_t)slice * 64 + lane;
  const int npair = width >> 1;
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
    const double idt = F.idt, mhnu = -0.5 * F.nu;
    double bf[GDIM];
#pragma unroll
    for (int d = 0; d < GDIM; ++d) bf[d] = 0.0;
    const unsigned short *__restrict__ mc =
        DICT ? reinterpret_cast<const unsigned short *>(F.Mc + base) + lane : nullptr;
    const unsigned short *__restrict__ kc =
        DICT ? reinterpret_cast<const unsigned short *>(F.Kc + base) + lane : nullptr;
    for (int k = 0; k < npair; ++k) {
      double2 m, kk;
      if constexpr (DICT) {
        const unsigned cm = mc[(size_t)k * 64], ck = kc[(size_t)k * 64];
        m.x = dM[cm & 0xff], m.y = dM[cm >> 8];
        kk.x = dK[ck & 0xff], kk.y = dK[ck >> 8];
      } else {
        m = mv[(size_t)k * 64], kk = kv[(size_t)k * 64];
      }
      const int2 col = cp[(size_t)k * 64];
      const double c0 = acc[(2 * k) * 64 + lane], c1 = acc[(2 * k + 1) * 64 + lane];
      // A = -0.5 C; A += (1/dt) M; A += (-0.5 nu) K        (fracstep.py:438-442)
      const double ar0 = fma(mhnu, kk.x, fma(idt, m.x, -0.5 * c0));
      const double ar1 = fma(mhnu, kk.y, fma(idt, m.y, -0.5 * c1));
      const double *x0 = F.u1 + (size_t)col.x * GDIM, *x1 = F.u1 + (size_t)col.y * GDIM;
#pragma unroll
      for (int d = 0; d < GDIM; ++d) bf[d] = fma(ar1, x1[d], fma(ar0, x0[d], bf[d]));
      // A = -A; A += (2/dt) M                               (fracstep.py:468-469)
      double2 v;
      v.x = fma(2.0 * idt, m.x, -ar0);
      v.y = fma(2.0 * idt, m.y, -ar1);
      av[(size_t)k * 64] = v;
    }
    if (row < A.n_rows) {
#pragma unroll
      for (int d = 0; d < GDIM; ++d)
        F.b_first[row * GDIM + d] = bf[d] + F.b0[row * GDIM + d];  // + b0 (fracstep.py:456)
    }
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
  auto kern = k_assemble_rows<GDIM, DEG, KIND, PW, DICT>;
  for (int b = 0; b < n_bins; ++b) {
    const int64_t cnt = bin_ptr[b + 1] - bin_ptr[b];
    if (cnt <= 0) continue;
    const size_t lds = (size_t)bin_width[b] * 64 * sizeof(double);
    if (lds > 160 * 1024) OX_FAIL("assemble: row width %d needs %zu B of LDS (> 160 KiB)", bin_width[b], lds);
    if (lds > 64 * 1024)
      OX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(64), lds, st, *cells, cell_dofs, *adj, adj_pos,
                       *A, F, bin_slices + bin_ptr[b]);
    OX_LAUNCH_CHECK();
  }
  return 0;
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
  OX_ROWS_CASE(2, 1, 4) OX_ROWS_CASE(2, 2, 8) OX_ROWS_CASE(3, 1, 4) OX_ROWS_CASE(3, 2, 16)
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

extern "C" int ox_assemble_first(int degree, const ox_cells *cells, const int32_t *cell_dofs,
                                 const ox_adj *adj, const uint8_t *adj_pos, int pw, const ox_sell *A,
                                 const ox_sell *M, const ox_sell *K, const double *uab,
                                 const double *u1, const double *b0, double *b_first, double dt,
                                 double nu, int n_bins, const int64_t *bin_ptr_host,
                                 const int32_t *bin_slices, const int32_t *bin_width_host,
                                 void *stream) {
  if (!cells || !cell_dofs || !adj || !adj_pos || !A || !M || !K || !M->vals || !K->vals || !uab || !u1 ||
      !b0 || !b_first)
    OX_FAIL("ox_assemble_first: null argument");
  if (M->slice_ptr != A->slice_ptr || K->slice_ptr != A->slice_ptr)
    OX_FAIL("ox_assemble_first: M, K and A must share one sparsity pattern");
  if (!(dt > 0.0)) OX_FAIL("ox_assemble_first: dt=%g", dt);
  FirstArgs F{M->vals, K->vals, uab, u1, b0, b_first, 1.0 / dt, nu,
              M->vcode, K->vcode, M->vdict, K->vdict, M->n_dict, K->n_dict};
  if (ox_prof_on) ox_prof_start(OX_TAG_ASSEMBLE_FIRST, ox_stream(stream));
  const int rc = launch_rows<OX_KIND_CONV>(degree, cells, cell_dofs, adj, adj_pos, pw, A, F, n_bins,
                                           bin_ptr_host, bin_slices, bin_width_host, ox_stream(stream));
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
  OX_W_CASE(2, 1) OX_W_CASE(2, 2) OX_W_CASE(3, 1) OX_W_CASE(3, 2)
#undef OX_W_CASE
  OX_FAIL("ox_assemble_weights: unsupported gdim=%d degree=%d", g, degree);
}

// kind 0: out[r][d] = base[r][d] + scale * int p * d_d(phi_r)       (A6, fracstep.py:487-497)
// kind 1: out[r][d] = base[r][d] + scale * int d_d(p) * phi_r       (A8, fracstep.py:618)
template <int GDIM, int RDEG, int PDEG, int KIND>
__global__ __launch_bounds__(256) void k_grad_vector(ox_cells cells, const int32_t *__restrict__ cell_pdofs,
                                                     ox_adj adj, int64_t n_rows,
                                                     const double *__restrict__ p,
                                                     const double *__restrict__ base_v, double scale,
                                                     double *__restrict__ out) {
  using ER = Elem<GDIM, RDEG>;
  using EP = Elem<GDIM, PDEG>;
  const auto &R = rt<GDIM, RDEG>();
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
  OX_GV_ALL(2) OX_GV_ALL(3)
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
  using ER = Elem<GDIM, RDEG>;
  using EU = Elem<GDIM, UDEG>;
  const auto &R = rt<GDIM, RDEG>();
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
  OX_DV_CASE(2, 1, 1) OX_DV_CASE(2, 1, 2) OX_DV_CASE(3, 1, 1) OX_DV_CASE(3, 1, 2)
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
  using ER = Elem<GDIM, RDEG>;
  using EC = Elem<GDIM, CDEG>;
  const auto &R = rt<GDIM, RDEG>();
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
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
#undef OX_RECT_DIM
#undef OX_RECT_CASE
  OX_FAIL("ox_assemble_rect: unsupported gdim=%d row_degree=%d col_degree=%d family=%d", g, row_degree,
          col_degree, family);
}
