/*
 * TEST INFRASTRUCTURE ONLY -- plain C (+OpenMP) restatement of the per-time-step IPCS hot
 * path of oasisx's FractionalStep_AB_CN, in CSR, one velocity component at a time, exactly as
 * the reference drives DOLFINx/PETSc.  Used (a) by tests/ as a second, compiled checker that is
 * validated against oracle/ipcs_oracle.py, (b) by bench.py's cpu_baseline leg as the timed host
 * baseline ("kind": "port").  Nothing under oasisx_amd/ links or loads this file.
 *
 * Parity status: see the header of ipcs_oracle.py (pinned by analytic solutions, exactness
 * identities and scipy; the reference's own tests hold no golden vectors for this path).
 *
 * Element integrals are contractions with reference tensors the caller supplies (computed in
 * ipcs_oracle.py with collapsed Gauss-Jacobi rules) -- a different evaluation scheme from the
 * quadrature loops of the HIP kernels:
 *   conv  Ce[i][j] = |J| sum_{k,a} (G[a].uab_k) Tc[k][a][j][i],  Tc = int phi_k dphi_j/dl_a phi_i
 *   pvdx  be[i][d] = |J| sum_a G[a][d] sum_c p_c Tp[i][c][a],    Tp = int psi_c dphi_i/dl_a
 *   gradp be[i][d] = |J| sum_a G[a][d] sum_c p_c Tg[i][c][a],    Tg = int dpsi_c/dl_a phi_i
 *   div   be[i]    = |J| sum_{k,a} (G[a].u_k) Td[i][k][a],       Td = int dphi_k/dl_a psi_i
 * Reference lines: fracstep.py:355-358 (conv), :306-309 (pvdx), :343-346 (gradp), :328-330 (div),
 * :432-472 (assemble_first), :508-525, :553-605, :607-658 (solves); ksp.py:71-78.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int cpu_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* y = A x (CSR) -- PETSc MatMult (fracstep.py:452,615,638) */
void cpu_spmv(int64_t n, const int64_t *rp, const int32_t *ci, const double *v, const double *x,
              double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) s += v[k] * x[ci[k]];
    y[r] = s;
  }
}

static double dot(int64_t n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

static int test_conv(double rn, double bn, double rtol, double atol) {
  if (!(rn == rn) || isinf(rn)) return -9;
  if (rn <= atol) return 3;
  if (rn <= rtol * bn) return 2;
  if (bn > 0.0 && rn >= 1e4 * bn) return -4; /* KSPConvergedDefault: -ksp_divtol, PETSc's default 1e4 */
  return 0;
}

/* Jacobi-preconditioned CG with PETSc's conventions (left PC, preconditioned residual norm,
 * rtol relative to ||D^-1 b||).  work: 4*n doubles. */
int cpu_cg(int64_t n, const int64_t *rp, const int32_t *ci, const double *v, const double *dinv,
           const double *b, double *x, double rtol, double atol, int max_it, int guess, double *work,
           int *its_out, double *rn_out) {
  double *r = work, *z = work + n, *p = work + 2 * n, *q = work + 3 * n;
  if (guess) {
    cpu_spmv(n, rp, ci, v, x, q);
  }
  double bn2 = 0.0, zz = 0.0, rz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : bn2, zz, rz)
  for (int64_t i = 0; i < n; ++i) {
    double ri = b[i];
    if (guess) ri -= q[i];
    else x[i] = 0.0;
    const double zi = dinv[i] * ri, db = dinv[i] * b[i];
    r[i] = ri;
    z[i] = zi;
    p[i] = zi;
    bn2 += db * db;
    zz += zi * zi;
    rz += ri * zi;
  }
  const double bn = sqrt(bn2);
  double rn = sqrt(zz);
  int it = 0, reason = test_conv(rn, bn, rtol, atol);
  while (!reason) {
    cpu_spmv(n, rp, ci, v, p, q);
    const double pq = dot(n, p, q);
    if (pq == 0.0) { reason = -5; break; }
    const double alpha = rz / pq;
    double rzn = 0.0;
    zz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : zz, rzn)
    for (int64_t i = 0; i < n; ++i) {
      x[i] += alpha * p[i];
      const double ri = r[i] - alpha * q[i];
      const double zi = dinv[i] * ri;
      r[i] = ri;
      z[i] = zi;
      zz += zi * zi;
      rzn += ri * zi;
    }
    rn = sqrt(zz);
    ++it;
    reason = test_conv(rn, bn, rtol, atol);
    if (reason) break;
    if (it >= max_it) { reason = -3; break; }
    const double beta = rzn / rz;
    rz = rzn;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
  }
  *its_out = it;
  *rn_out = rn;
  return reason;
}

/* Left-Jacobi-preconditioned BiCGStab (PETSc KSPBCGS conventions).  work: 6*n doubles. */
int cpu_bicgstab(int64_t n, const int64_t *rp, const int32_t *ci, const double *v,
                 const double *dinv, const double *b, double *x, double rtol, double atol,
                 int max_it, int guess, double *work, int *its_out, double *rn_out) {
  double *r = work, *rh = work + n, *p = work + 2 * n, *vv = work + 3 * n, *s = work + 4 * n,
         *t = work + 5 * n;
  if (guess) cpu_spmv(n, rp, ci, v, x, t);
  double bn2 = 0.0, rr = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : bn2, rr)
  for (int64_t i = 0; i < n; ++i) {
    double ri = b[i];
    if (guess) ri -= t[i];
    else x[i] = 0.0;
    ri *= dinv[i];
    const double db = dinv[i] * b[i];
    r[i] = ri;
    rh[i] = ri;
    p[i] = 0.0;
    vv[i] = 0.0;
    bn2 += db * db;
    rr += ri * ri;
  }
  const double bn = sqrt(bn2);
  double rn = sqrt(rr);
  int it = 0, reason = test_conv(rn, bn, rtol, atol);
  double rho = 1.0, alpha = 1.0, omega = 1.0, rho_new = rr;
  while (!reason) {
    if (rho_new == 0.0) { reason = -5; break; }
    const double beta = (rho_new / rho) * (alpha / omega);
    rho = rho_new;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * vv[i]);
    cpu_spmv(n, rp, ci, v, p, vv);
    double rv = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rv)
    for (int64_t i = 0; i < n; ++i) {
      vv[i] *= dinv[i];
      rv += rh[i] * vv[i];
    }
    if (rv == 0.0) { reason = -5; break; }
    alpha = rho / rv;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) s[i] = r[i] - alpha * vv[i];
    cpu_spmv(n, rp, ci, v, s, t);
    double tt = 0.0, ts = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : tt, ts)
    for (int64_t i = 0; i < n; ++i) {
      t[i] *= dinv[i];
      tt += t[i] * t[i];
      ts += t[i] * s[i];
    }
    omega = tt != 0.0 ? ts / tt : 0.0;
    rr = 0.0;
    rho_new = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rr, rho_new)
    for (int64_t i = 0; i < n; ++i) {
      x[i] += alpha * p[i] + omega * s[i];
      const double ri = s[i] - omega * t[i];
      r[i] = ri;
      rr += ri * ri;
      rho_new += rh[i] * ri;
    }
    rn = sqrt(rr);
    ++it;
    reason = test_conv(rn, bn, rtol, atol);
    if (reason) break;
    if (omega == 0.0) { reason = -5; break; }
    if (it >= max_it) { reason = -3; break; }
  }
  *its_out = it;
  *rn_out = rn;
  return reason;
}

/* position of column c in the sorted row r */
static inline int64_t csr_find(const int64_t *rp, const int32_t *ci, int64_t r, int32_t c) {
  int64_t lo = rp[r], hi = rp[r + 1] - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ci[mid] < c) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

/* geom[c]: (d+1) x d barycentric gradients then |detJ|  (stride (d+1)*d+1) */
#define MAXND 20 /* Lagrange P3 on tetrahedra: 4 + 12 + 4 */
#define MAXV 4

/* C = assemble_matrix(inner(dot(uab, nabla_grad(u)), v)*dx), cell loop + scatter-add
 * (fracstep.py:435-437).  uab is SoA [d][n].  Cv must be zeroed by the caller. */
void cpu_assemble_convection(int d, int nd, int64_t ncells, const double *geom,
                             const int32_t *cell_dofs, const double *Tc, int64_t n,
                             const double *uab, const int64_t *rp, const int32_t *ci, double *Cv) {
  const int nv = d + 1, gs = nv * d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < ncells; ++c) {
    const double *G = geom + c * gs;
    const double adet = G[nv * d];
    const int32_t *dd = cell_dofs + c * nd;
    double w[MAXND][MAXV];
    for (int k = 0; k < nd; ++k)
      for (int a = 0; a < nv; ++a) {
        double s = 0.0;
        for (int e = 0; e < d; ++e) s += G[a * d + e] * uab[(int64_t)e * n + dd[k]];
        w[k][a] = s;
      }
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j) {
        double s = 0.0;
        for (int k = 0; k < nd; ++k)
          for (int a = 0; a < nv; ++a) s += w[k][a] * Tc[((k * nv + a) * nd + j) * nd + i];
        const int64_t pos = csr_find(rp, ci, dd[i], dd[j]);
        const double val = adet * s;
#pragma omp atomic
        Cv[pos] += val;
      }
  }
}

/* kind 0: out[d][r] += int p d_d(phi_r) ; kind 1: out[d][r] += int d_d(p) phi_r  (T = Tp or Tg) */
void cpu_assemble_grad_vector(int d, int nd_row, int nd_p, int64_t ncells, const double *geom,
                              const int32_t *cell_rdofs, const int32_t *cell_pdofs, const double *T,
                              const double *p, int64_t n_rows, double *out) {
  const int nv = d + 1, gs = nv * d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < ncells; ++c) {
    const double *G = geom + c * gs;
    const double adet = G[nv * d];
    const int32_t *rd = cell_rdofs + c * nd_row, *pd = cell_pdofs + c * nd_p;
    for (int i = 0; i < nd_row; ++i) {
      double S[MAXV];
      for (int a = 0; a < nv; ++a) {
        double s = 0.0;
        for (int k = 0; k < nd_p; ++k) s += p[pd[k]] * T[(i * nd_p + k) * nv + a];
        S[a] = s;
      }
      for (int e = 0; e < d; ++e) {
        double s = 0.0;
        for (int a = 0; a < nv; ++a) s += S[a] * G[a * d + e];
        const double val = adet * s;
#pragma omp atomic
        out[(int64_t)e * n_rows + rd[i]] += val;
      }
    }
  }
}

/* out[r] += int div(u) psi_r, u SoA [d][n_u] */
void cpu_assemble_div_vector(int d, int nd_row, int nd_u, int64_t ncells, const double *geom,
                             const int32_t *cell_rdofs, const int32_t *cell_udofs, const double *Td,
                             int64_t n_u, const double *u, double *out) {
  const int nv = d + 1, gs = nv * d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < ncells; ++c) {
    const double *G = geom + c * gs;
    const double adet = G[nv * d];
    const int32_t *rd = cell_rdofs + c * nd_row, *ud = cell_udofs + c * nd_u;
    double w[MAXND][MAXV];
    for (int k = 0; k < nd_u; ++k)
      for (int a = 0; a < nv; ++a) {
        double s = 0.0;
        for (int e = 0; e < d; ++e) s += G[a * d + e] * u[(int64_t)e * n_u + ud[k]];
        w[k][a] = s;
      }
    for (int i = 0; i < nd_row; ++i) {
      double s = 0.0;
      for (int k = 0; k < nd_u; ++k)
        for (int a = 0; a < nv; ++a) s += w[k][a] * Td[(i * nd_u + k) * nv + a];
      const double val = adet * s;
#pragma omp atomic
      out[rd[i]] += val;
    }
  }
}

/* The S3 sequence of assemble_first on same-pattern value arrays (fracstep.py:438-442,468-469):
 * phase 0: A = -0.5*C + (1/dt)*M - 0.5*nu*K   (C given in A on entry)
 * phase 1: A = -A + (2/dt)*M */
void cpu_matrix_phase(int phase, int64_t nnz, double *A, const double *M, const double *K, double dt,
                      double nu) {
  if (phase == 0) {
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nnz; ++k) {
      double a = -0.5 * A[k];
      a += (1.0 / dt) * M[k];
      a += (-0.5 * nu) * K[k];
      A[k] = a;
    }
  } else {
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nnz; ++k) A[k] = -A[k] + (2.0 / dt) * M[k];
  }
}

/* Mat.zeroRowsLocal(rows, 1.0) (fracstep.py:471-472) */
void cpu_zero_rows(const int64_t *rp, const int32_t *ci, double *v, const int32_t *rows, int64_t nrows) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nrows; ++i) {
    const int32_t r = rows[i];
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) v[k] = (ci[k] == r) ? 1.0 : 0.0;
  }
}

void cpu_diag_inv(int64_t n, const int64_t *rp, const int32_t *ci, const double *v, double *dinv) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double dg = 1.0;
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k)
      if (ci[k] == r) dg = v[k];
    dinv[r] = dg != 0.0 ? 1.0 / dg : 1.0;
  }
}

void cpu_axpby(int64_t n, double a, const double *x, double b, const double *y, double *z) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) z[i] = a * x[i] + (b != 0.0 ? b * y[i] : 0.0);
}

/* ---- own set-up of the port (so that a full-size cross-check shares NOTHING with the product but
 * the mesh definition): dof -> cell adjacency, CSR pattern, mass and stiffness assembly ---------- */

/* adj_ptr[r] .. adj_ptr[r+1]: the cells that contain dof r (ascending), from a cell->dof table */
void cpu_adjacency(int64_t n_rows, int64_t ncells, int nd, const int32_t *cell_dofs, int64_t *adj_ptr,
                   int32_t *adj_cells) {
  memset(adj_ptr, 0, sizeof(int64_t) * (size_t)(n_rows + 1));
  for (int64_t i = 0; i < ncells * nd; ++i) adj_ptr[cell_dofs[i] + 1]++;
  for (int64_t r = 0; r < n_rows; ++r) adj_ptr[r + 1] += adj_ptr[r];
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_rows + 1));
  memcpy(fill, adj_ptr, sizeof(int64_t) * (size_t)(n_rows + 1));
  for (int64_t c = 0; c < ncells; ++c)
    for (int k = 0; k < nd; ++k) adj_cells[fill[cell_dofs[c * nd + k]]++] = (int32_t)c;
  free(fill);
}

static int cmp_i32(const void *a, const void *b) {
  const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return (x > y) - (x < y);
}

/* CSR pattern of the operator with rows = dofs of the adjacency's space and columns = the col_dofs
 * of the adjacent cells (DOLFINx create_matrix / create_sparsity_pattern, fracstep.py:293-300,324).
 * ci == NULL: count only (fills rp); otherwise fill ci (sorted rows) with the rp of the first call. */
void cpu_pattern(int64_t n_rows, int64_t n_cols, const int64_t *adj_ptr, const int32_t *adj_cells, int nd_c,
                 const int32_t *col_dofs, int64_t *rp, int32_t *ci) {
  if (!ci) rp[0] = 0;
#pragma omp parallel
  {
    /* thread-private marker: mark[c] == r + 1 <=> column c already seen in row r */
    int64_t *mark = (int64_t *)calloc((size_t)n_cols, sizeof(int64_t));
    int32_t *buf = NULL;
    int64_t cap = 0;
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
      const int64_t nc = adj_ptr[r + 1] - adj_ptr[r];
      if (nc * nd_c > cap) {
        cap = 2 * nc * nd_c;
        buf = (int32_t *)realloc(buf, sizeof(int32_t) * (size_t)cap);
      }
      int64_t u = 0;
      for (int64_t a = adj_ptr[r]; a < adj_ptr[r + 1]; ++a)
        for (int k = 0; k < nd_c; ++k) {
          const int32_t c = col_dofs[(int64_t)adj_cells[a] * nd_c + k];
          if (mark[c] != r + 1) {
            mark[c] = r + 1;
            buf[u++] = c;
          }
        }
      if (ci) {
        qsort(buf, (size_t)u, sizeof(int32_t), cmp_i32);
        memcpy(ci + rp[r], buf, sizeof(int32_t) * (size_t)u);
      } else {
        rp[r + 1] = u;
      }
    }
    free(buf);
    free(mark);
  }
  if (!ci)
    for (int64_t r = 0; r < n_rows; ++r) rp[r + 1] += rp[r];
}

/* kind 0: u*v*dx with Tm[i][j] = int phi_i phi_j;  kind 1: inner(grad u, grad v)*dx with
 * Tk[i][j][a][b] = int dphi_i/dl_a dphi_j/dl_b (reference simplex), Ae = |J| sum_ab (G[a].G[b]) Tk.
 * assemble_matrix of fracstep.py:373-380; vals must be zeroed by the caller. */
void cpu_assemble_matrix(int kind, int d, int nd, int64_t ncells, const double *geom,
                         const int32_t *cell_dofs, const double *T, const int64_t *rp,
                         const int32_t *ci, double *vals) {
  const int nv = d + 1, gs = nv * d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < ncells; ++c) {
    const double *G = geom + c * gs;
    const double adet = G[nv * d];
    const int32_t *dd = cell_dofs + c * nd;
    double gg[MAXV][MAXV];
    if (kind == 1)
      for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b) {
          double s = 0.0;
          for (int e = 0; e < d; ++e) s += G[a * d + e] * G[b * d + e];
          gg[a][b] = s;
        }
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j) {
        double s;
        if (kind == 0) {
          s = T[i * nd + j];
        } else {
          s = 0.0;
          for (int a = 0; a < nv; ++a)
            for (int b = 0; b < nv; ++b) s += gg[a][b] * T[((i * nd + j) * nv + a) * nv + b];
        }
        const int64_t pos = csr_find(rp, ci, dd[i], dd[j]);
        const double val = adet * s;
#pragma omp atomic
        vals[pos] += val;
      }
  }
}

/* w[r] += int phi_r dx  (Ti[i] = int phi_i on the reference simplex) */
void cpu_assemble_weights(int d, int nd, int64_t ncells, const double *geom, const int32_t *cell_dofs,
                          const double *Ti, double *w) {
  const int gs = (d + 1) * d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < ncells; ++c)
    for (int i = 0; i < nd; ++i) {
      const double val = geom[c * gs + (d + 1) * d] * Ti[i];
#pragma omp atomic
      w[cell_dofs[c * nd + i]] += val;
    }
}

void cpu_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n < 1 ? 1 : n);
#else
  (void)n;
#endif
}
