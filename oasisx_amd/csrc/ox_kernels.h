// Internal launch helpers shared by the translation units of liboasisx_hip.so.
#pragma once
#include "ox_common.h"

// fused SpMV epilogues
#define OX_EPI_NONE 0    // y = A x
#define OX_EPI_DOT 1     // + partial[c] = sum x_row * y_row            (CG: p.q)
#define OX_EPI_BCGS_V 2  // y = D^-1 A x, partial[c] = sum aux_row*y_row (BiCGStab: rhat.v)
#define OX_EPI_BCGS_T 3  // y = D^-1 A x, partial = {y.y, y.x_row}       (BiCGStab: t.t, t.s)
#define OX_EPI_BCGS_T5 4 // y = D^-1 A x, partial = {y.y, y.x, aux.x, aux.y, x.x}  (merged-reduction BiCGStab:
                         //   t.t, t.s, rhat.s, rhat.t, s.s -- omega, rho and |r| from ONE reduction)
#define OX_EPI_CG_M2 5   // y = A x, partial = {x.y, y.(D^-1 y)}   (merged-reduction CG: p.q and q.D^-1 q give alpha AND, by
                         // r' = r - alpha q, the next r.z -- one synchronisation point; D^-1 through ox_spmv_set_epilogue_dinv)
__host__ __device__ constexpr int ox_epi_nv(int epi, int nc) {
  return epi == OX_EPI_NONE ? 0 : (epi == OX_EPI_BCGS_T ? 2 * nc : (epi == OX_EPI_BCGS_T5 ? 5 * nc : (epi == OX_EPI_CG_M2 ? 2 * nc : nc)));
}

#define OX_VEC_MAX_BLOCKS 4096  // upper bound of the grid cap of the BLAS-1 kernels (sizes the partial arrays)
// The cap itself depends on the vector: measured on one box with the bench workload (OX_VEC_BLOCKS = 512 / 768 /
// 1024 / 1536 / 2048): the BiCGStab solve on the 3 x 17 M-row vectors takes 44.0 / 42.1 / 40.8 / 41.2 / 44.3 ms per
// step (fewer, longer blocks keep fewer DRAM streams open), the pressure-CG iteration on 2.1 M rows 69.8 / 68.3 /
// 68.1 / 67.1 / 67.0 us (8 short blocks per CU hide the launch ramp).  OX_VEC_BLOCKS overrides (tuning).
#include <stdlib.h>
static inline int ox_vec_cap(int64_t n) {
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("OX_VEC_BLOCKS");
    env = e ? atoi(e) : 0;
    if (env > OX_VEC_MAX_BLOCKS) env = OX_VEC_MAX_BLOCKS;
  }
  if (env >= 8) return env;
  return n >= ((int64_t)1 << 24) ? 1024 : 2048;
}

#define OX_SPMV_MAX_BLOCKS (1 << 22)  // one slice group per block (a cap of 2048 = persistent grid
                                      // measured 10 % slower: r01 notes in DESIGN.md)
#define OX_MAX_NV 15            // most sums reduced at one synchronisation point (5 * OX_MAXC: merged-reduction BiCGStab)

static inline int ox_spmv_blocks_n(int n_slices) {
  const int ngroups = (n_slices + 3) / 4;
  if (ngroups == 0) return 0;
  const int g8 = (ngroups + 7) & ~7;
  return g8 > OX_SPMV_MAX_BLOCKS ? OX_SPMV_MAX_BLOCKS : g8;
}
static inline int ox_spmv_blocks(const ox_sell *A) { return ox_spmv_blocks_n(A->n_slices); }

static inline int ox_vec_blocks(int64_t n) {
  int64_t b = (n / 2 + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > ox_vec_cap(n) ? ox_vec_cap(n) : b);
}

#define OX_RED_THREADS 1024  // widest final-reduction block (many partials: velocity SpMV)
#define OX_RED_THREADS_SMALL 256
static inline int ox_red_threads(int nparts) { return nparts > 4096 ? OX_RED_THREADS : OX_RED_THREADS_SMALL; }

// Per-thread slice of the ordered final reduction: v[i] = sum over this thread's partials
// (4 partial rows in flight per thread).  blockDim.x threads take part.
__device__ __forceinline__ void ox_gather_partials(const double *__restrict__ partial, int nparts,
                                                   int nv, double (&v)[OX_MAX_NV]) {
  const int T = blockDim.x;
#pragma unroll
  for (int i = 0; i < OX_MAX_NV; ++i) v[i] = 0.0;
  int p = threadIdx.x;
  for (; p + 3 * T < nparts; p += 4 * T) {
    double t[4][OX_MAX_NV];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < OX_MAX_NV; ++i)
        t[u][i] = (i < nv) ? partial[(size_t)(p + u * T) * nv + i] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < OX_MAX_NV; ++i) v[i] += t[u][i];
  }
  for (; p < nparts; p += T) {
#pragma unroll
    for (int i = 0; i < OX_MAX_NV; ++i)
      if (i < nv) v[i] += partial[(size_t)p * nv + i];
  }
}

// Sum the first nv of the per-thread values over the block (blockDim.x = 64 * nwaves <= 1024);
// the result is valid in thread 0.  Only nv slots are reduced (nv is wave-uniform).
__device__ __forceinline__ void ox_block_sum_wide(double (&v)[OX_MAX_NV], int nv, double *lds /* [16*OX_MAX_NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < OX_MAX_NV; ++i) {
    if (i < nv) {
      const double s = ox_wave_sum(v[i]);
      if (lane == 0) lds[i * 16 + wave] = s;
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < OX_MAX_NV; ++i) {
      if (i < nv) {
        double s = lane < nw ? lds[i * 16 + lane] : 0.0;
        // fixed-order tree over <= 16 wave sums
        s += __shfl_down(s, 8, 64);
        s += __shfl_down(s, 4, 64);
        s += __shfl_down(s, 2, 64);
        s += __shfl_down(s, 1, 64);
        v[i] = s;
      }
    }
  }
}

int ox_spmv_launch(const ox_sell *A, const double *x, double *y, int ncomp, int epi,
                   const double *dinv, const double *aux, double *partial, const int *done,
                   hipStream_t st);
// distributed mat-vec (halo exchange overlapped with the interior slices where the operator carries the
// interior / boundary split) and the number of per-block partials its fused epilogue writes
int ox_spmv_dist(const ox_sell *A, double *x, double *y, int ncomp, int epi, const double *dinv, const double *aux,
                 double *partial, const int *done, const ox_dist *dist, hipStream_t st);
int ox_spmv_dist_nparts(const ox_sell *A, const ox_dist *dist, int ncomp);
// optional 1-byte codes + dictionary of the diagonal the OX_EPI_CG_M2 epilogue multiplies by (nullptr: the f64 array)
struct OxEpiDinv {
  const uint8_t *code;
  const double *dict;
};
void ox_spmv_set_epilogue_dinv(const uint8_t *code, const double *dict);
int ox_reduce_partials(const double *partial, int nparts, int nv, double *sums, hipStream_t st);

// ---- optional per-kernel HIP-event timing (bench.py: roofline.achieved is measured live, on the
// stream the kernel runs on) --------------------------------------------------------------------
#define OX_TAG_SPMV(nc, epi) (10 * (nc) + (epi))
#define OX_TAG_ASSEMBLE_FIRST 100
#define OX_TAG_GRAD_VECTOR 110
#define OX_TAG_DIV_VECTOR 120
#define OX_TAG_ASSEMBLE_MATRIX 130
#define OX_TAG_RECT_S2V 140
#define OX_TAG_RECT_V2S 141
#define OX_TAG_HALO 150       // one halo exchange (key = components); includes waiting for the peers
#define OX_TAG_SYNC_POINT 151 // one distributed Krylov synchronisation point (reduce + all-reduce + logic)
extern bool ox_prof_on;
void ox_prof_start(int tag, hipStream_t st, long long key = 0);
void ox_prof_stop(hipStream_t st);
