// Internal launch helpers shared by the translation units of liboasisx_hip.so.
#pragma once
#include "ox_common.h"

// fused SpMV epilogues
#define OX_EPI_NONE 0    // y = A x
#define OX_EPI_DOT 1     // + partial[c] = sum x_row * y_row            (CG: p.q)
#define OX_EPI_BCGS_V 2  // y = D^-1 A x, partial[c] = sum aux_row*y_row (BiCGStab: rhat.v)
#define OX_EPI_BCGS_T 3  // y = D^-1 A x, partial = {y.y, y.x_row}       (BiCGStab: t.t, t.s)

#define OX_VEC_MAX_BLOCKS 2048  // grid cap of the BLAS-1 kernels (256 CUs x 8 blocks)

static inline int ox_spmv_blocks(const ox_sell *A) { return (A->n_slices + 3) / 4; }
static inline int ox_vec_blocks(int64_t n) {
  int64_t b = (n / 2 + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > OX_VEC_MAX_BLOCKS ? OX_VEC_MAX_BLOCKS : b);
}

int ox_spmv_launch(const ox_sell *A, const double *x, double *y, int ncomp, int epi,
                   const double *dinv, const double *aux, double *partial, const int *done,
                   hipStream_t st);
int ox_reduce_partials(const double *partial, int nparts, int nv, double *sums, hipStream_t st);

// ---- optional per-kernel HIP-event timing (bench.py: roofline.achieved is measured live, on the
// stream the kernel runs on) --------------------------------------------------------------------
#define OX_TAG_SPMV(nc, epi) (10 * (nc) + (epi))
#define OX_TAG_ASSEMBLE_FIRST 100
#define OX_TAG_GRAD_VECTOR 110
#define OX_TAG_DIV_VECTOR 120
#define OX_TAG_ASSEMBLE_MATRIX 130
extern bool ox_prof_on;
void ox_prof_start(int tag, hipStream_t st);
void ox_prof_stop(hipStream_t st);
