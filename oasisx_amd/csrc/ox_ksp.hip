// Jacobi-preconditioned CG and BiCGStab on up to 3 right-hand sides that share one
// SELL-64 matrix (reference ksp.py:71-78 -> PETSc KSPSolve; fracstep.py:521,578,634).
//
// All scalars (alpha, beta, omega, rho, norms, convergence flags) live in a device-side
// state block; the host only enqueues kernels and reads the state back every
// `check_every` iterations.  Once every component has converged the state's `done` flag
// turns the remaining queued kernels into no-ops, so the result is exactly the one an
// every-iteration check would give.  Reductions are two-stage and ordered (no float
// atomics): results are bit-reproducible run to run.
#include "ox_kernels.h"
#include "ox_ksp_dev.h"
#include "ox_p2p.h"
#include <type_traits>

// Fused: reduce per-block partials (fixed order) + scalar logic.  One wide block.
// second partial array of a synchronisation point (the single-reduction CG merges the sums of the
// update kernel and of the SpMV): its nv2 sums follow the first array's nv
struct KspPart2 {
  const double *partial;
  int nparts, nv;
};

__device__ __forceinline__ void ksp_gather2(const double *__restrict__ partial, int nparts, int nv, const KspPart2 &B,
                                            double (&v)[OX_MAX_NV]) {
  ox_gather_partials(partial, nparts, nv, v);
  if (B.nv > 0) {
    double w[OX_MAX_NV];
    ox_gather_partials(B.partial, B.nparts, B.nv, w);
#pragma unroll
    for (int i = 0; i < OX_MAX_NV; ++i)
#pragma unroll
      for (int j = 0; j < OX_MAX_NV; ++j)
        if (j == i - nv && j < B.nv) v[i] = w[j];
  }
}


// (register arrays sized by the phase: ksp_ph_nv -- OX_MAX_NV-wide ones spill in this 1024-thread block)
template <int NVT>
__device__ __forceinline__ void ksp_gather_t(const double *__restrict__ partial, int nparts, int nv, double (&v)[NVT]) {
  const int T = blockDim.x;
#pragma unroll
  for (int i = 0; i < NVT; ++i) v[i] = 0.0;
  int p = threadIdx.x;
  for (; p + 3 * T < nparts; p += 4 * T) {
    double t[4][NVT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NVT; ++i) t[u][i] = (i < nv) ? partial[(size_t)(p + u * T) * nv + i] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NVT; ++i) v[i] += t[u][i];
  }
  for (; p < nparts; p += T) {
#pragma unroll
    for (int i = 0; i < NVT; ++i)
      if (i < nv) v[i] += partial[(size_t)p * nv + i];
  }
}
// v[0..NV) = this thread's share of `nparts` partial rows of exactly NV sums, up to U rows per thread requested together
// (thread t takes rows t, t + T, ...: the order the generic gather sums in)
template <int NV, int U, int NVT>
__device__ __forceinline__ void ksp_gather_rows(const double *__restrict__ partial, int nparts, double (&v)[NVT]) {
  const int T = blockDim.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  for (int p0 = threadIdx.x; p0 < nparts; p0 += U * T) {
    double t[U][NV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * T;
#pragma unroll
      for (int i = 0; i < NV; ++i) t[u][i] = p < nparts ? partial[(size_t)p * NV + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += t[u][i];
  }
}
template <int NVT>
__device__ __forceinline__ void ksp_block_sum_t(double (&v)[NVT], int nv, double *lds /* [16 * NVT] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NVT; ++i) {
    if (i < nv) {
      const double s = ox_wave_sum(v[i]);
      if (lane == 0) lds[i * 16 + wave] = s;
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
      if (i < nv) {
        double s = lane < nw ? lds[i * 16 + lane] : 0.0;
        s += __shfl_down(s, 8, 64);
        s += __shfl_down(s, 4, 64);
        s += __shfl_down(s, 2, 64);
        s += __shfl_down(s, 1, 64);
        v[i] = s;
      }
    }
  }
}

// The sums of one synchronisation point: this thread's share of the partial rows (and of the point's second array B),
// then the block tree.  Register arrays sized by the phase (NVT = ksp_ph_nv * columns): OX_MAX_NV-wide ones spill in a
// 1024-thread block (16 us instead of 5).  Returns the number of sums; they are valid in thread 0.
template <int PH, int NVT>
__device__ __forceinline__ int ksp_point_sums(const double *__restrict__ partial, int nparts, int nv, const KspPart2 &B,
                                              double (&v)[NVT], double *red /* [16 * NVT] */) {
  if constexpr (PH == PH_CGM_IT) {
    // 8 392 x 2 partial sums of the pressure mat-vec at 128^3: every thread's rows requested in ONE round (the
    // generic gather keeps 4 rows in flight)
    ksp_gather_rows<2, 10>(partial, nparts, v);
    v[2] = v[3] = v[4] = 0.0;
  } else if (nv == 1) {  // one right-hand side: a thread's <= 10 rows requested in ONE round (same order of sums as the
    ksp_gather_rows<1, 10>(partial, nparts, v);  // generic gather, which keeps 4 rows in flight: 2-3 rounds)
#pragma unroll
    for (int i = 1; i < NVT; ++i) v[i] = 0.0;
  } else if (nv == 2 && NVT >= 2) {
    ksp_gather_rows<(NVT >= 2 ? 2 : 1), 10>(partial, nparts, v);
#pragma unroll
    for (int i = 2; i < NVT; ++i) v[i] = 0.0;
  } else {
    ksp_gather_t<NVT>(partial, nparts, nv, v);
  }
  if (B.nv > 0) {  // second partial array of the point (single-reduction CG): its sums follow the first's
    double w[NVT];
    ksp_gather_t<NVT>(B.partial, B.nparts, B.nv, w);
#pragma unroll
    for (int i = 0; i < NVT; ++i)
#pragma unroll
      for (int j = 0; j < NVT; ++j)
        if (j == i - nv && j < B.nv) v[i] = w[j];
  }
  nv += B.nv;
  ksp_block_sum_t<NVT>(v, nv, red);  // contains a __syncthreads()
  return nv;
}
// (the merged-reduction CG is a one-column solver: 3 + 2 sums)
template <int PH>
constexpr int ksp_nvt() { return PH == PH_CGM_IT ? ksp_ph_nv(PH) : ksp_ph_nv(PH) * OX_MAXC; }

template <int PH>
__global__ __launch_bounds__(OX_RED_THREADS) void k_ksp_scalar(KspState *S,
                                                               const double *__restrict__ partial,
                                                               int nparts, int nv, KspParams P, KspPart2 B) {
  constexpr int NVT = ksp_nvt<PH>();
  __shared__ double red[16 * NVT];
  __shared__ double sums[NVT];
  __shared__ KspState sh;
  // state and partials are loaded in ONE memory round trip; the done flag is looked at afterwards
  // (a leading `if (S->done) return` costs a dependent round trip of its own, ~1 us per launch)
  ksp_state_load(&sh, S);
  double v[NVT];
  ksp_point_sums<PH, NVT>(partial, nparts, nv, B, v, red);  // contains a __syncthreads(): sh is complete after it
  if (!ksp_is_init(PH) && sh.done) {  // uniform: the state stays as it is; the host's copy of it is still due
    if (P.mirror) ksp_state_store(P.mirror, &sh);
    return;
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NVT; ++i) sums[i] = v[i];  // (the logic indexes the sums at run time: from LDS, not scratch)
    for (int c = 0; c < P.nc; ++c) ksp_logic<PH>(&sh, sums, c, P);
    ksp_finish(&sh, P.nc_total);
  }
  __syncthreads();
  ksp_state_store(S, &sh);
  if (P.mirror) ksp_state_store(P.mirror, &sh);  // the host reads the state of a batch from here: no copy kernel in the stream
}

// Reduction only (RCCL plans: the rank's sums of the point, both partial arrays in ONE launch, for the all-reduce that
// follows).  Round 5: the generic k_reduce_partials -- OX_MAX_NV-wide arrays in a 1024-thread block, one launch per
// array -- took 17.4 + 9.5 us of a 110-us partitioned pressure iteration (rocprofv3 on the self-loop plan of rank 0 of
// 8 at 256^3: tools/predict_scaling.py); this is k_ksp_scalar's own gather.
template <int PH>
__global__ __launch_bounds__(OX_RED_THREADS) void k_ksp_reduce(const double *__restrict__ partial, int nparts, int nv, KspPart2 B,
                                                               double *__restrict__ out) {
  constexpr int NVT = ksp_nvt<PH>();
  __shared__ double red[16 * NVT];
  double v[NVT];
  const int n = ksp_point_sums<PH, NVT>(partial, nparts, nv, B, v, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NVT; ++i)
      if (i < n) out[i] = v[i];
  }
}

// The same on the direct xGMI transport: the block's sums are all-reduced over the ranks' windows
// inside the kernel (rank order: identical bits, hence identical decisions, on every rank), so a
// distributed synchronisation point stays ONE kernel.
template <int PH>
__global__ __launch_bounds__(OX_RED_THREADS) void k_ksp_scalar_p2p(KspState *S,
                                                                   const double *__restrict__ partial,
                                                                   int nparts, int nv, KspParams P, ox_p2p_ar ar,
                                                                   KspPart2 B) {
  constexpr int NVT = ksp_nvt<PH>();
  __shared__ double red[16 * NVT];
  __shared__ KspState sh;
  __shared__ double vals[OX_P2P_MAXV + 1];
  __shared__ double stage[64][OX_P2P_MAXV + 1];
  ksp_state_load(&sh, S);
  double v[NVT];
  nv = ksp_point_sums<PH, NVT>(partial, nparts, nv, B, v, red);  // contains a __syncthreads(): sh is complete after it
  // A queued sync point that runs after `done` still takes part in the exchange (the host has advanced
  // the sequence number for it; skipping would let two LIVE exchanges share a parity slot), it only
  // leaves the state alone.  `done` is the same on every rank (rank-ordered sums: identical bits).
  const bool idle = !ksp_is_init(PH) && sh.done;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NVT; ++i)
      if (i < nv) vals[i] = idle ? 0.0 : v[i];
  }
  __syncthreads();
  ox_p2p_allreduce_block(vals, nv, ar, stage);
  if (idle) {
    if (P.mirror) ksp_state_store(P.mirror, &sh);
    return;
  }
  if (threadIdx.x == 0) {  // (the logic indexes the sums at run time: `vals` is in LDS)
    for (int c = 0; c < P.nc; ++c) ksp_logic<PH>(&sh, vals, c, P);
    ksp_finish(&sh, P.nc_total);
  }
  __syncthreads();
  ksp_state_store(S, &sh);
  if (P.mirror) ksp_state_store(P.mirror, &sh);
}

// Logic only (distributed runs: the sums were all-reduced over the ranks first).
template <int PH>
__global__ __launch_bounds__(64) void k_ksp_logic(KspState *S, const double *__restrict__ sums, KspParams P) {
  __shared__ KspState sh;
  __shared__ double sv[OX_MAX_NV];
  ksp_state_load(&sh, S);
  if (threadIdx.x < OX_MAX_NV) sv[threadIdx.x] = sums[threadIdx.x];
  __syncthreads();
  if (!ksp_is_init(PH) && sh.done) {
    if (P.mirror) ksp_state_store(P.mirror, &sh);
    return;
  }
  if (threadIdx.x == 0) {
    for (int c = 0; c < P.nc; ++c) ksp_logic<PH>(&sh, sv, c, P);
    ksp_finish(&sh, P.nc_total);
  }
  __syncthreads();
  ksp_state_store(S, &sh);
  if (P.mirror) ksp_state_store(P.mirror, &sh);
}

// ------------------------------- vector kernels ------------------------------------------
// thread = row, NC interleaved components per row (24-B contiguous per lane for NC = 3).
#define OX_ROW_LOOP                                 \
  const int64_t stride_ = (int64_t)gridDim.x * 256; \
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += stride_)

template <int NV>
__device__ __forceinline__ void ksp_store_partial(double (&s)[NV], double *red, double *partial) {
  ox_block_sum_256<NV>(s, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) partial[(size_t)blockIdx.x * NV + i] = s[i];
  }
}

// Flat traversal of an interleaved (n x NC) block: every thread takes TWO consecutive elements, so
// loads and stores are 16 B wide and a wave touches every cache line once (a thread-per-row loop
// issues NC 8-byte accesses at a 24-byte stride: each of them sweeps all lines of the wave's span,
// which held the 3-component kernels at ~4 TB/s where the 1-component ones reach 5.6).
// f(e, c0, r0, c1, r1, two): elements e (column c0, row r0) and e+1 (c1, r1); two = false for the
// odd tail element.
template <int NC, class F>
__device__ __forceinline__ void ox_flat_pairs(int64_t n, F &&f) {
  const int64_t tot = n * NC, n2 = tot >> 1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
    const int64_t e = 2 * i;
    const int64_t r0 = e / NC;
    const int c0 = (int)(e - r0 * NC);
    const bool wrap = (c0 + 1 == NC);
    f(e, c0, r0, wrap ? 0 : c0 + 1, wrap ? r0 + 1 : r0, true);
  }
  if ((tot & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t e = tot - 1, r0 = e / NC;
    f(e, (int)(e - r0 * NC), r0, 0, r0, false);
  }
}
__device__ __forceinline__ double2 ox_ld2(const double *p, int64_t e, bool two) {
  if (two) return *reinterpret_cast<const double2 *>(p + e);
  double2 v;
  v.x = p[e];
  v.y = 0.0;
  return v;
}
__device__ __forceinline__ void ox_st2(double *p, int64_t e, double2 v, bool two) {
  if (two) *reinterpret_cast<double2 *>(p + e) = v;
  else p[e] = v.x;
}
template <int NC>
__device__ __forceinline__ double ox_sel(const double (&a)[NC], int c) {
  double v = a[0];
#pragma unroll
  for (int k = 1; k < NC; ++k) v = (c == k) ? a[k] : v;
  return v;
}
// s[off + c] = fma(a, b, s[off + c]) with a runtime column c
template <int NC, int NV>
__device__ __forceinline__ void ox_acc(double (&s)[NV], int off, int c, double a, double b) {
#pragma unroll
  for (int k = 0; k < NC; ++k) s[off + k] = (c == k) ? fma(a, b, s[off + k]) : s[off + k];
}

// CG init: r = b - q (q = A x0) or r = b, x = 0;  z = D^-1 r (not stored);  p = z
// partial = {r.z, z.z, (D^-1 b).(D^-1 b)}
template <int NC>
__global__ __launch_bounds__(256) void k_cg_init(int64_t n, const double *__restrict__ b, double *x,
                                                 const double *q, const double *__restrict__ dinv,
                                                 double *vr, double *vp, int guess,
                                                 double *partial) {
  __shared__ double red[4 * 3 * NC];
  double s[3 * NC];
#pragma unroll
  for (int i = 0; i < 3 * NC; ++i) s[i] = 0.0;
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t ra, int cb, int64_t rb, bool two) {  // (see k_bcgs_init)
    const double2 bb = ox_ld2(b, e, two);
    double2 r = bb;
    if (guess) {
      const double2 qq = ox_ld2(q, e, two);
      r.x -= qq.x;
      r.y -= qq.y;
    } else {
      double2 z0;
      z0.x = z0.y = 0.0;
      ox_st2(x, e, z0, two);
    }
    const double d0 = dinv[ra], d1 = two ? dinv[rb] : 0.0;
    double2 z;
    z.x = d0 * r.x;
    z.y = d1 * r.y;
    const double db0 = d0 * bb.x, db1 = d1 * bb.y;
    ox_st2(vr, e, r, two);
    ox_st2(vp, e, z, two);
    ox_acc<NC>(s, 0, ca, r.x, z.x);
    ox_acc<NC>(s, NC, ca, z.x, z.x);
    ox_acc<NC>(s, 2 * NC, ca, db0, db0);
    if (two) {
      ox_acc<NC>(s, 0, cb, r.y, z.y);
      ox_acc<NC>(s, NC, cb, z.y, z.y);
      ox_acc<NC>(s, 2 * NC, cb, db1, db1);
    }
  });
  ksp_store_partial<3 * NC>(s, red, partial);
}

// The Jacobi diagonal of the CG vector kernels.  A matrix assembled on congruent cells has few distinct diagonal
// values (Ap on the box meshes: 4, M: 8): with a dictionary of <= 256 values the two update kernels read ONE
// byte per row instead of eight -- 2 of the 10 vector passes of a pressure-CG iteration (the kernels run at
// ~6 TB/s: bandwidth is their bound).  Same values, same operations: bit-identical iterates.
struct KspDinv {
  const double *v;      // [n_rows]
  const uint8_t *code;  // [n_rows] or nullptr
  const double *dict;   // [n]
  int n;
};
__device__ __forceinline__ void ksp_dinv_stage(const KspDinv &D, double *dd /* LDS [256] */) {
  if (D.code) {  // uniform
    for (int i = threadIdx.x; i < D.n; i += blockDim.x) dd[i] = D.dict[i];
    __syncthreads();
  }
}
#define KSP_DINV(row) (D.code ? dd[D.code[row]] : D.v[row])

// CG, first vector kernel of an iteration: r -= alpha q; z = D^-1 r (not stored); partial = {r.z, z.z}.
// x waits for the second kernel, which reads p anyway: 10 vector passes per iteration instead of 11,
// no z vector; every element sees the same operations as in the textbook order.
template <int NC>
__global__ __launch_bounds__(256) void k_cg_update1(int64_t n, const KspState *S, int c0, double *vr,
                                                    const double *__restrict__ vq, KspDinv D,
                                                    double *partial) {
  __shared__ double red[4 * 2 * NC];
  __shared__ double dd[256];
  if (S->done) return;
  ksp_dinv_stage(D, dd);
  double alpha[NC], s[2 * NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) alpha[c] = S->alpha[c0 + c];
#pragma unroll
  for (int i = 0; i < 2 * NC; ++i) s[i] = 0.0;
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t ra, int cb, int64_t rb, bool two) {
    const double2 qq = ox_ld2(vq, e, two);
    double2 r = ox_ld2(vr, e, two);
    const double a0 = ox_sel<NC>(alpha, ca), a1 = ox_sel<NC>(alpha, cb);
    const double d0 = KSP_DINV(ra), d1 = two ? KSP_DINV(rb) : 0.0;
    r.x = fma(-a0, qq.x, r.x);
    r.y = fma(-a1, qq.y, r.y);
    double2 z;
    z.x = d0 * r.x;
    z.y = d1 * r.y;
    ox_st2(vr, e, r, two);
    ox_acc<NC>(s, 0, ca, r.x, z.x);
    ox_acc<NC>(s, NC, ca, z.x, z.x);
    if (two) {
      ox_acc<NC>(s, 0, cb, r.y, z.y);
      ox_acc<NC>(s, NC, cb, z.y, z.y);
    }
  });
  ksp_store_partial<2 * NC>(s, red, partial);
}

// CG, second vector kernel: x += alpha p (the iteration's alpha, still in the state); p = D^-1 r + beta p.
// When the test after the first kernel ends the solve this kernel is skipped like every queued one:
// the host then runs it once with `finish` set so that x receives its last update.
template <int NC>
__global__ __launch_bounds__(256) void k_cg_update2(int64_t n, const KspState *S, int c0, double *x,
                                                    const double *__restrict__ vr, KspDinv D, double *vp, int finish) {
  __shared__ double dd[256];
  if (!finish && S->done) return;
  ksp_dinv_stage(D, dd);
  double alpha[NC], beta[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    alpha[c] = S->alpha[c0 + c];
    beta[c] = S->beta[c0 + c];
  }
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t ra, int cb, int64_t rb, bool two) {
    const double2 r = ox_ld2(vr, e, two);
    double2 p = ox_ld2(vp, e, two), xx = ox_ld2(x, e, two);
    const double d0 = KSP_DINV(ra), d1 = two ? KSP_DINV(rb) : 0.0;
    xx.x = fma(ox_sel<NC>(alpha, ca), p.x, xx.x);
    xx.y = fma(ox_sel<NC>(alpha, cb), p.y, xx.y);
    const double z0 = d0 * r.x, z1 = d1 * r.y;
    p.x = fma(ox_sel<NC>(beta, ca), p.x, z0);
    p.y = fma(ox_sel<NC>(beta, cb), p.y, z1);
    ox_st2(x, e, xx, two);
    ox_st2(vp, e, p, two);
  });
}

// ---------------------------------------------------------------------------------------
// One-column CG on one GPU with the two synchronisation points FOLDED into the update kernels (round 4).
// A pressure-CG iteration at 128^3 was five kernels -- mat-vec 31 us, k_ksp_scalar 4.7, k_cg_update1 10.8,
// k_ksp_scalar 4.7, k_cg_update2 12.9 --: a third of the two scalar kernels' time is the kernel boundary, the rest two
// dependent round trips of ONE block while 255 CUs idle.  Here every block of the consuming update kernel sums the
// producer's partial sums itself (same per-thread rows, same block tree as k_ksp_scalar with 1024 threads: the same bits
// in every block), runs the point's scalar logic on an LDS copy of the state, and goes on to its rows; block 0 stores the
// new state.  Few large blocks (<= OX_CG_FOLD_BLOCKS x 1024 threads) keep the redundant reads small (8 392 sums x 512
// blocks = 34 MB of L2 hits; with 2 048 blocks of 256 threads -- the first attempt of the round, profiles/
// r04_krylov_consumer_fold_experiment.txt -- they cost more than they saved).  The state ping-pongs between two blocks
// (a block that starts late must not read what block 0 has already written): S -> S2 at the first point, S2 -> S at the
// second, so that between iterations it is where every other kernel and the host expect it.
// ---------------------------------------------------------------------------------------
#define OX_FOLD_T 1024
#define OX_FOLD_NP 5  // pairs of rows per thread whose operands are requested BEFORE the point's reduction
// LDS-only barrier: waits for this wave's LDS traffic, not for its outstanding global loads (__syncthreads() fences
// both, which would park the prefetched operands' latency in front of the reduction instead of behind it)
__device__ __forceinline__ void ox_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The point itself.  Loads are issued in this order: state, partial sums, dictionary -- and only then (by the caller's
// `prefetch`) the operands of the thread's rows, so that waiting for the former does not wait for the latter (vector
// loads return in order).  Every load of the prologue is unconditional (clamped address + select): a load inside a
// branch makes the compiler wait for ALL outstanding loads where the branches join.
template <int NVIN, int U, int PH, bool CODE, class Prefetch>
__device__ __forceinline__ bool ksp_fold_point(const KspState *Sin, KspState *Sout, const double *__restrict__ pin, int npin,
                                               const KspParams &P, const KspDinv &D, KspState *sh, double *red /* [32] */,
                                               double *sums /* [2] */, double *dd /* [256] */, Prefetch &&prefetch) {
  const int T = OX_FOLD_T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
  const unsigned long long sw = reinterpret_cast<const unsigned long long *>(Sin)[min(tid, KSP_STATE_WORDS - 1)];
  double t[U][NVIN];  // thread t takes rows t, t + T, ...: the order k_ksp_scalar sums in
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int pc = min(tid + u * T, npin - 1);
#pragma unroll
    for (int i = 0; i < NVIN; ++i) t[u][i] = pin[(size_t)pc * NVIN + i];
  }
  double dv = 0.0;
  if constexpr (CODE) dv = D.dict[min(tid, D.n - 1)];
  prefetch();
  // (a fence the compiler cannot schedule across: every load above is issued before it -- "memory" --, and every use of
  // the point's operands depends on its output; left alone the selects below were hoisted in front of the rows' loads,
  // which then waited behind the partial sums' round trip: k_cg_update1f, seen in the ISA)
  int np_ = npin;
  asm volatile("" : "+s"(np_) : : "memory");
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int i = 0; i < NVIN; ++i) t[u][i] = tid + u * T < np_ ? t[u][i] : 0.0;
  // (unconditional: a load used only inside a branch is sunk into it, behind the prefetch; the surplus threads rewrite
  // the last word with its own value)
  reinterpret_cast<unsigned long long *>(sh)[min(tid, KSP_STATE_WORDS - 1)] = sw;
  if constexpr (CODE) dd[min(tid, D.n - 1)] = dv;  // (same reason)
  double v[2] = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int i = 0; i < NVIN; ++i) v[i] += t[u][i];
  if (np_ > U * T) {  // (more than U x 1024 sums: not on this path's sizes)
    for (int p = tid + U * T; p < np_; p += T)
#pragma unroll
      for (int i = 0; i < NVIN; ++i) v[i] += pin[(size_t)p * NVIN + i];
  }
#pragma unroll
  for (int i = 0; i < NVIN; ++i) {
    const double s = ox_wave_sum(v[i]);
    if (lane == 0) red[i * 16 + wave] = s;
  }
  ox_lds_barrier();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NVIN; ++i) {
      double s = lane < nw ? red[i * 16 + lane] : 0.0;
      s += __shfl_down(s, 8, 64);
      s += __shfl_down(s, 4, 64);
      s += __shfl_down(s, 2, 64);
      s += __shfl_down(s, 1, 64);
      v[i] = s;
    }
    if (tid == 0 && !sh->done) {
      sums[0] = v[0];
      sums[1] = v[1];
      for (int c = 0; c < P.nc; ++c) ksp_logic<PH>(sh, sums, c, P);
      ksp_finish(sh, P.nc_total);
    }
  }
  ox_lds_barrier();
  // (a state that arrived `done` is handed on unchanged -- the logic above did not run --; the host's copy is still due)
  if (blockIdx.x == 0) {
    ksp_state_store(Sout, sh);
    if (P.mirror) ksp_state_store(P.mirror, sh);
  }
  return !sh->done;
}

// first point {p.q} -> alpha;  r -= alpha q;  z = D^-1 r (not stored);  partial = {r.z, z.z}   (k_cg_update1<1>)
// (n >= 2: the prefetch reads clamped addresses)
template <bool CODE>
__global__ __launch_bounds__(OX_FOLD_T) void k_cg_update1f(int64_t n, const KspState *Sin, KspState *Sout,
                                                           const double *__restrict__ pin, int npin, KspParams P,
                                                           double *vr, const double *__restrict__ vq, KspDinv D,
                                                           double *pout) {
  __shared__ double red[32], sums[2], dd[256];
  __shared__ KspState sh;
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * OX_FOLD_T, i0 = (int64_t)blockIdx.x * OX_FOLD_T + threadIdx.x;
  double2 qq[OX_FOLD_NP], rr[OX_FOLD_NP], dq[OX_FOLD_NP];
  unsigned short cq[OX_FOLD_NP];
  const bool live = ksp_fold_point<1, 10, PH_CG_A, CODE>(Sin, Sout, pin, npin, P, D, &sh, red, sums, dd, [&]() {
#pragma unroll
    for (int j = 0; j < OX_FOLD_NP; ++j) {
      const int64_t iw = i0 + j * stride;
      const int64_t i = iw < n2 ? iw : min(i0, n2 - 1);  // (surplus: the thread's own first pair again, not ONE hot address)
      qq[j] = *reinterpret_cast<const double2 *>(vq + 2 * i);
      rr[j] = *reinterpret_cast<const double2 *>(vr + 2 * i);
      if constexpr (CODE) cq[j] = *reinterpret_cast<const unsigned short *>(D.code + 2 * i);
      else dq[j] = *reinterpret_cast<const double2 *>(D.v + 2 * i);
    }
  });
  if (!live) return;  // (uniform)
  const double alpha = sh.alpha[P.c0];
  double s[2] = {0.0, 0.0};
  auto rows = [&](int64_t e, const double2 q2, double2 r, const double d0, const double d1) {
    r.x = fma(-alpha, q2.x, r.x);
    r.y = fma(-alpha, q2.y, r.y);
    const double z0 = d0 * r.x, z1 = d1 * r.y;
    *reinterpret_cast<double2 *>(vr + e) = r;
    s[0] = fma(r.x, z0, s[0]);
    s[1] = fma(z0, z0, s[1]);
    s[0] = fma(r.y, z1, s[0]);
    s[1] = fma(z1, z1, s[1]);
  };
#pragma unroll
  for (int j = 0; j < OX_FOLD_NP; ++j) {
    const int64_t i = i0 + j * stride;
    if (i < n2) rows(2 * i, qq[j], rr[j], CODE ? dd[cq[j] & 0xff] : dq[j].x, CODE ? dd[cq[j] >> 8] : dq[j].y);
  }
  for (int64_t i = i0 + OX_FOLD_NP * stride; i < n2; i += stride) {
    const int64_t e = 2 * i;
    rows(e, *reinterpret_cast<const double2 *>(vq + e), *reinterpret_cast<const double2 *>(vr + e), KSP_DINV(e), KSP_DINV(e + 1));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t e = n - 1;
    const double r = fma(-alpha, vq[e], vr[e]), z = KSP_DINV(e) * r;
    vr[e] = r;
    s[0] = fma(r, z, s[0]);
    s[1] = fma(z, z, s[1]);
  }
  __syncthreads();  // (red is reused)
  ksp_block_sum_t<2>(s, 2, red);
  if (threadIdx.x == 0) {
    pout[(size_t)blockIdx.x * 2] = s[0];
    pout[(size_t)blockIdx.x * 2 + 1] = s[1];
  }
}

// second point {r.z, z.z} -> convergence test, beta;  x += alpha p;  p = D^-1 r + beta p   (k_cg_update2<1>; when the
// test ends the solve the rows are left alone and the host runs k_cg_update2<1> with `finish` set, as in the unfolded form)
template <bool CODE>
__global__ __launch_bounds__(OX_FOLD_T) void k_cg_update2f(int64_t n, const KspState *Sin, KspState *Sout,
                                                           const double *__restrict__ pin, int npin, KspParams P,
                                                           double *x, const double *__restrict__ vr, KspDinv D, double *vp) {
  __shared__ double red[32], sums[2], dd[256];
  __shared__ KspState sh;
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * OX_FOLD_T, i0 = (int64_t)blockIdx.x * OX_FOLD_T + threadIdx.x;
  double2 rr[OX_FOLD_NP], pp[OX_FOLD_NP], xx[OX_FOLD_NP], dq[OX_FOLD_NP];
  unsigned short cq[OX_FOLD_NP];
  const bool live = ksp_fold_point<2, 1, PH_CG_B, CODE>(Sin, Sout, pin, npin, P, D, &sh, red, sums, dd, [&]() {
#pragma unroll
    for (int j = 0; j < OX_FOLD_NP; ++j) {
      const int64_t iw = i0 + j * stride;
      const int64_t i = iw < n2 ? iw : min(i0, n2 - 1);  // (surplus: the thread's own first pair again, not ONE hot address)
      rr[j] = *reinterpret_cast<const double2 *>(vr + 2 * i);
      pp[j] = *reinterpret_cast<const double2 *>(vp + 2 * i);
      xx[j] = *reinterpret_cast<const double2 *>(x + 2 * i);
      if constexpr (CODE) cq[j] = *reinterpret_cast<const unsigned short *>(D.code + 2 * i);
      else dq[j] = *reinterpret_cast<const double2 *>(D.v + 2 * i);
    }
  });
  if (!live) return;
  const double alpha = sh.alpha[P.c0], beta = sh.beta[P.c0];
  auto rows = [&](int64_t e, const double2 r, double2 p, double2 xv, const double d0, const double d1) {
    xv.x = fma(alpha, p.x, xv.x);
    xv.y = fma(alpha, p.y, xv.y);
    p.x = fma(beta, p.x, d0 * r.x);
    p.y = fma(beta, p.y, d1 * r.y);
    *reinterpret_cast<double2 *>(x + e) = xv;
    *reinterpret_cast<double2 *>(vp + e) = p;
  };
#pragma unroll
  for (int j = 0; j < OX_FOLD_NP; ++j) {
    const int64_t i = i0 + j * stride;
    if (i < n2) rows(2 * i, rr[j], pp[j], xx[j], CODE ? dd[cq[j] & 0xff] : dq[j].x, CODE ? dd[cq[j] >> 8] : dq[j].y);
  }
  for (int64_t i = i0 + OX_FOLD_NP * stride; i < n2; i += stride) {
    const int64_t e = 2 * i;
    rows(e, *reinterpret_cast<const double2 *>(vr + e), *reinterpret_cast<const double2 *>(vp + e),
         *reinterpret_cast<const double2 *>(x + e), KSP_DINV(e), KSP_DINV(e + 1));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t e = n - 1;
    const double pe = vp[e];
    x[e] = fma(alpha, pe, x[e]);
    vp[e] = fma(beta, pe, KSP_DINV(e) * vr[e]);
  }
}

// Merged-reduction CG, the ONE vector kernel of an iteration (alpha and beta are both known behind the mat-vec's
// synchronisation point, PH_CGM_IT):  x += alpha p;  r -= alpha q;  z = D^-1 r (not stored);  p = z + beta p;
// partial = {r.z, z.z, p.q} -- the true sums, which replace the carried r.z / serve the convergence test at the next
// point, and the new direction against the OLD q (= q_next . p_old, A being symmetric: see PH_CGM_IT).  7 vector passes
// where k_cg_update1 + k_cg_update2 make 8, one launch and one synchronisation point less per iteration.
template <int NC>
__global__ __launch_bounds__(256) void k_cgm_update(int64_t n, const KspState *S, int c0, double *x, double *vr,
                                                    const double *__restrict__ vq, KspDinv D, double *vp,
                                                    double *partial) {
  __shared__ double red[4 * 3 * NC];
  __shared__ double dd[256];
  if (S->done) return;
  ksp_dinv_stage(D, dd);
  double alpha[NC], beta[NC], s[3 * NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    alpha[c] = S->alpha[c0 + c];
    beta[c] = S->beta[c0 + c];
  }
#pragma unroll
  for (int i = 0; i < 3 * NC; ++i) s[i] = 0.0;
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t ra, int cb, int64_t rb, bool two) {
    const double2 qq = ox_ld2(vq, e, two);
    double2 r = ox_ld2(vr, e, two), p = ox_ld2(vp, e, two), xx = ox_ld2(x, e, two);
    const double a0 = ox_sel<NC>(alpha, ca), a1 = ox_sel<NC>(alpha, cb);
    const double d0 = KSP_DINV(ra), d1 = two ? KSP_DINV(rb) : 0.0;
    xx.x = fma(a0, p.x, xx.x);
    xx.y = fma(a1, p.y, xx.y);
    r.x = fma(-a0, qq.x, r.x);
    r.y = fma(-a1, qq.y, r.y);
    const double z0 = d0 * r.x, z1 = d1 * r.y;
    p.x = fma(ox_sel<NC>(beta, ca), p.x, z0);
    p.y = fma(ox_sel<NC>(beta, cb), p.y, z1);
    ox_st2(x, e, xx, two);
    ox_st2(vr, e, r, two);
    ox_st2(vp, e, p, two);
    ox_acc<NC>(s, 0, ca, r.x, z0);
    ox_acc<NC>(s, NC, ca, z0, z0);
    ox_acc<NC>(s, 2 * NC, ca, p.x, qq.x);
    if (two) {
      ox_acc<NC>(s, 0, cb, r.y, z1);
      ox_acc<NC>(s, NC, cb, z1, z1);
      ox_acc<NC>(s, 2 * NC, cb, p.y, qq.y);
    }
  });
  ksp_store_partial<3 * NC>(s, red, partial);
}

// Single-reduction CG (Chronopoulos & Gear) init: r = b - q (q = A x0) or r = b, x = 0; u = D^-1 r;
// p = s = 0;  partial = {r.u, u.u, (D^-1 b).(D^-1 b)}
template <int NC>
__global__ __launch_bounds__(256) void k_cgs_init(int64_t n, const double *__restrict__ b, double *x,
                                                  const double *q, const double *__restrict__ dinv,
                                                  double *vr, double *vu, double *vp, double *vs, int guess,
                                                  double *partial) {
  __shared__ double red[4 * 3 * NC];
  double s[3 * NC];
#pragma unroll
  for (int i = 0; i < 3 * NC; ++i) s[i] = 0.0;
  OX_ROW_LOOP {
    const double d = dinv[row];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t i = row * NC + c;
      const double bi = b[i];
      double ri = bi;
      if (guess) ri -= q[i];
      else x[i] = 0.0;
      const double ui = d * ri, db = d * bi;
      vr[i] = ri;
      vu[i] = ui;
      vp[i] = 0.0;
      vs[i] = 0.0;
      s[c] = fma(ri, ui, s[c]);
      s[NC + c] = fma(ui, ui, s[NC + c]);
      s[2 * NC + c] = fma(db, db, s[2 * NC + c]);
    }
  }
  ksp_store_partial<3 * NC>(s, red, partial);
}

// Single-reduction CG, the whole vector update of an iteration in one pass:
//   p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  u = D^-1 r;   partial = {r.u, u.u}
// (u_old = D^-1 r_old is recomputed, not read: 6 reads + 5 writes per entry, what the two update
// kernels of the standard recurrences move together)
template <int NC>
__global__ __launch_bounds__(256) void k_cgs_update(int64_t n, const KspState *S, int c0, double *x, double *vr,
                                                    double *vu, double *vp, double *vs,
                                                    const double *__restrict__ vw,
                                                    const double *__restrict__ dinv, double *partial) {
  __shared__ double red[4 * 2 * NC];
  if (S->done) return;
  double alpha[NC], beta[NC], s[2 * NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    alpha[c] = S->alpha[c0 + c];
    beta[c] = S->beta[c0 + c];
  }
#pragma unroll
  for (int i = 0; i < 2 * NC; ++i) s[i] = 0.0;
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t ra, int cb, int64_t rb, bool two) {
    const double2 w = ox_ld2(vw, e, two);
    double2 p = ox_ld2(vp, e, two), ss = ox_ld2(vs, e, two), xx = ox_ld2(x, e, two), r = ox_ld2(vr, e, two);
    const double a0 = ox_sel<NC>(alpha, ca), a1 = ox_sel<NC>(alpha, cb);
    const double b0 = ox_sel<NC>(beta, ca), b1 = ox_sel<NC>(beta, cb);
    const double d0 = dinv[ra], d1 = two ? dinv[rb] : 0.0;
    p.x = fma(b0, p.x, d0 * r.x);
    p.y = fma(b1, p.y, d1 * r.y);
    ss.x = fma(b0, ss.x, w.x);
    ss.y = fma(b1, ss.y, w.y);
    xx.x = fma(a0, p.x, xx.x);
    xx.y = fma(a1, p.y, xx.y);
    r.x = fma(-a0, ss.x, r.x);
    r.y = fma(-a1, ss.y, r.y);
    double2 u;
    u.x = d0 * r.x;
    u.y = d1 * r.y;
    ox_st2(vp, e, p, two);
    ox_st2(vs, e, ss, two);
    ox_st2(x, e, xx, two);
    ox_st2(vr, e, r, two);
    ox_st2(vu, e, u, two);
    ox_acc<NC>(s, 0, ca, r.x, u.x);
    ox_acc<NC>(s, NC, ca, u.x, u.x);
    if (two) {
      ox_acc<NC>(s, 0, cb, r.y, u.y);
      ox_acc<NC>(s, NC, cb, u.y, u.y);
    }
  });
  ksp_store_partial<2 * NC>(s, red, partial);
}

// BiCGStab init: r = D^-1 (b - q) or D^-1 b (x = 0); rhat = r; p = r.
// (The textbook starts from p = v = 0 and forms p = r + beta (p - omega v) = r in its first iteration -- exactly r,
// bit for bit -- so the first k_bcgs_p is skipped and v, which the first mat-vec writes before anyone reads it, is
// not initialised: 5 vector passes less per solve.)
// partial = {r.r, (D^-1 b).(D^-1 b)}
template <int NC>
__global__ __launch_bounds__(256) void k_bcgs_init(int64_t n, const double *__restrict__ b,
                                                   double *x, const double *q,
                                                   const double *__restrict__ dinv, double *vr,
                                                   double *vrhat, double *vp, double *vv, int guess,
                                                   double *partial) {
  __shared__ double red[4 * 2 * NC];
  double s[2 * NC];
#pragma unroll
  for (int i = 0; i < 2 * NC; ++i) s[i] = 0.0;
  // (flat 16-byte traversal like the update kernels: the row loop's 8-byte accesses at a 24-byte stride ran this
  // kernel at 2.6 TB/s on three columns; the same operations per element)
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t ra, int cb, int64_t rb, bool two) {
    const double2 bb = ox_ld2(b, e, two);
    double2 r = bb;
    if (guess) {
      const double2 qq = ox_ld2(q, e, two);
      r.x -= qq.x;
      r.y -= qq.y;
    } else {
      double2 z;
      z.x = z.y = 0.0;
      ox_st2(x, e, z, two);
    }
    const double d0 = dinv[ra], d1 = two ? dinv[rb] : 0.0;
    r.x *= d0;
    r.y *= d1;
    const double db0 = d0 * bb.x, db1 = d1 * bb.y;
    ox_st2(vr, e, r, two);
    ox_st2(vrhat, e, r, two);
    ox_st2(vp, e, r, two);
    ox_acc<NC>(s, 0, ca, r.x, r.x);
    ox_acc<NC>(s, NC, ca, db0, db0);
    if (two) {
      ox_acc<NC>(s, 0, cb, r.y, r.y);
      ox_acc<NC>(s, NC, cb, db1, db1);
    }
  });
  ksp_store_partial<2 * NC>(s, red, partial);
}

// BiCGStab: p = r + beta (p - omega v)   (or, on a restart: rhat <- r, p <- r)
template <int NC>
__global__ __launch_bounds__(256) void k_bcgs_p(int64_t n, const KspState *S, int c0,
                                                const double *__restrict__ vr, double *vp,
                                                const double *__restrict__ vv, double *vrhat) {
  if (S->done) return;
  double beta[NC], omega[NC], reseed[NC];
  bool any = false;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    beta[c] = S->beta[c0 + c];
    omega[c] = S->omega[c0 + c];
    reseed[c] = S->restart[c0 + c] ? 1.0 : 0.0;
    any = any || S->restart[c0 + c];
  }
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t, int cb, int64_t, bool two) {
    const double2 r = ox_ld2(vr, e, two), v = ox_ld2(vv, e, two);
    double2 p = ox_ld2(vp, e, two);
    const bool ra = ox_sel<NC>(reseed, ca) != 0.0, rb = ox_sel<NC>(reseed, cb) != 0.0;
    p.x = ra ? r.x : fma(ox_sel<NC>(beta, ca), fma(-ox_sel<NC>(omega, ca), v.x, p.x), r.x);
    p.y = rb ? r.y : fma(ox_sel<NC>(beta, cb), fma(-ox_sel<NC>(omega, cb), v.y, p.y), r.y);
    ox_st2(vp, e, p, two);
    if (any) {  // restart after a breakdown: rhat <- r in the re-seeded columns
      double2 h = ox_ld2(vrhat, e, two);
      h.x = ra ? r.x : h.x;
      h.y = rb ? r.y : h.y;
      ox_st2(vrhat, e, h, two);
    }
  });
}

// BiCGStab: s = r - alpha v
template <int NC>
__global__ __launch_bounds__(256) void k_bcgs_s(int64_t n, const KspState *S, int c0,
                                                const double *__restrict__ vr,
                                                const double *__restrict__ vv, double *vs) {
  if (S->done) return;
  double alpha[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) alpha[c] = S->alpha[c0 + c];
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t, int cb, int64_t, bool two) {
    const double2 r = ox_ld2(vr, e, two), v = ox_ld2(vv, e, two);
    double2 s;
    s.x = fma(-ox_sel<NC>(alpha, ca), v.x, r.x);
    s.y = fma(-ox_sel<NC>(alpha, cb), v.y, r.y);
    ox_st2(vs, e, s, two);
  });
}

// BiCGStab: x += alpha p + omega s; r = s - omega t; partial = {r.r, rhat.r}
template <int NC>
__global__ __launch_bounds__(256) void k_bcgs_x(int64_t n, const KspState *S, int c0, double *x, double *vr,
                                                const double *__restrict__ vrhat,
                                                const double *__restrict__ vp,
                                                const double *__restrict__ vs,
                                                const double *__restrict__ vt, double *partial,
                                                int finish /* merged variant: the x update the `done` flag skipped */) {
  __shared__ double red[4 * 2 * NC];
  if (!finish && S->done) return;
  double alpha[NC], omega[NC], s[2 * NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    alpha[c] = S->alpha[c0 + c];
    omega[c] = S->omega[c0 + c];
  }
#pragma unroll
  for (int i = 0; i < 2 * NC; ++i) s[i] = 0.0;
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t, int cb, int64_t, bool two) {
    const double2 si = ox_ld2(vs, e, two), p = ox_ld2(vp, e, two), t = ox_ld2(vt, e, two);
    const double2 h = ox_ld2(vrhat, e, two);
    double2 xx = ox_ld2(x, e, two), r;
    const double a0 = ox_sel<NC>(alpha, ca), a1 = ox_sel<NC>(alpha, cb);
    const double w0 = ox_sel<NC>(omega, ca), w1 = ox_sel<NC>(omega, cb);
    xx.x = fma(w0, si.x, fma(a0, p.x, xx.x));
    xx.y = fma(w1, si.y, fma(a1, p.y, xx.y));
    r.x = fma(-w0, t.x, si.x);
    r.y = fma(-w1, t.y, si.y);
    ox_st2(x, e, xx, two);
    ox_st2(vr, e, r, two);
    ox_acc<NC>(s, 0, ca, r.x, r.x);
    ox_acc<NC>(s, NC, ca, h.x, r.x);
    if (two) {
      ox_acc<NC>(s, 0, cb, r.y, r.y);
      ox_acc<NC>(s, NC, cb, h.y, r.y);
    }
  });
  ksp_store_partial<2 * NC>(s, red, partial);
}

// Merged-reduction BiCGStab: behind its second synchronisation point alpha, omega AND the next beta are known, so
// the x / r update of iteration k and the p update of iteration k+1 are one kernel:
//   x += alpha p + omega s;   r = s - omega t;   p = r + beta (p - omega v)      (re-seeded column: rhat = p = r)
// reads s, p, t, x, v, writes x, r, p: 8 vector passes where k_bcgs_x + k_bcgs_p make 11 (rhat is not read: the
// residual norm and rho come from the recurrence).  The same order of operations per element as the two kernels.
template <int NC>
__global__ __launch_bounds__(256) void k_bcgs_xp(int64_t n, const KspState *S, int c0, double *x, double *vr, double *vrhat,
                                                 double *vp, const double *__restrict__ vs, const double *__restrict__ vt,
                                                 const double *__restrict__ vv,
                                                 int finish /* the x update the `done` flag skipped */) {
  if (!finish && S->done) return;
  double alpha[NC], omega[NC], beta[NC], reseed[NC];
  bool any = false;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    alpha[c] = S->alpha[c0 + c];
    omega[c] = S->omega[c0 + c];
    beta[c] = S->beta[c0 + c];
    reseed[c] = S->restart[c0 + c] ? 1.0 : 0.0;
    any = any || S->restart[c0 + c];
  }
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t, int cb, int64_t, bool two) {
    const double2 si = ox_ld2(vs, e, two), t = ox_ld2(vt, e, two), v = ox_ld2(vv, e, two);
    double2 p = ox_ld2(vp, e, two), xx = ox_ld2(x, e, two), r;
    const double a0 = ox_sel<NC>(alpha, ca), a1 = ox_sel<NC>(alpha, cb);
    const double w0 = ox_sel<NC>(omega, ca), w1 = ox_sel<NC>(omega, cb);
    xx.x = fma(w0, si.x, fma(a0, p.x, xx.x));
    xx.y = fma(w1, si.y, fma(a1, p.y, xx.y));
    r.x = fma(-w0, t.x, si.x);
    r.y = fma(-w1, t.y, si.y);
    const bool ra = ox_sel<NC>(reseed, ca) != 0.0, rb = ox_sel<NC>(reseed, cb) != 0.0;
    p.x = ra ? r.x : fma(ox_sel<NC>(beta, ca), fma(-w0, v.x, p.x), r.x);
    p.y = rb ? r.y : fma(ox_sel<NC>(beta, cb), fma(-w1, v.y, p.y), r.y);
    ox_st2(x, e, xx, two);
    ox_st2(vr, e, r, two);
    ox_st2(vp, e, p, two);
    if (any) {
      double2 h = ox_ld2(vrhat, e, two);
      h.x = ra ? r.x : h.x;
      h.y = rb ? r.y : h.y;
      ox_st2(vrhat, e, h, two);
    }
  });
}

// Merged-reduction BiCGStab, once per solve: partial = {r.r} of the stored residual (PH_BCGSM_FIN re-tests on it the
// columns its recurrence norm declared converged).
template <int NC>
__global__ __launch_bounds__(256) void k_rr(int64_t n, const double *__restrict__ vr, double *partial) {
  __shared__ double red[4 * NC];
  double s[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) s[i] = 0.0;
  ox_flat_pairs<NC>(n, [&](int64_t e, int ca, int64_t, int cb, int64_t, bool two) {
    const double2 r = ox_ld2(vr, e, two);
    ox_acc<NC>(s, 0, ca, r.x, r.x);
    if (two) ox_acc<NC>(s, 0, cb, r.y, r.y);
  });
  ksp_store_partial<NC>(s, red, partial);
}

// ------------------------------- host driver ---------------------------------------------
static inline size_t ox_align(size_t v) { return (v + 255) & ~(size_t)255; }

// One block reads partials at ~45 GB/s: the 66 312 x 6 partials of a velocity SpMV (3.2 MB) kept the
// scalar kernel of that synchronisation point busy for 79 us.  From OX_PRERED_MIN doubles on, blocks of
// OX_PRERED_CHUNK partial rows are summed first (fixed order) and the scalar kernel reads their sums.
#define OX_PRERED_CHUNK 256
#define OX_PRERED_MIN 16384
#define OX_PRERED_OFFSET 64  // doubles of the sums region kept for the sums themselves
__global__ __launch_bounds__(256) void k_prereduce(const double *__restrict__ partial, int nparts, int nv,
                                                   double *__restrict__ out) {
  __shared__ double red[16 * OX_MAX_NV];
  const int r0 = blockIdx.x * OX_PRERED_CHUNK;
  const int cnt = min(OX_PRERED_CHUNK, nparts - r0);
  double v[OX_MAX_NV];
  ox_gather_partials(partial + (size_t)r0 * nv, cnt, nv, v);
  ox_block_sum_wide(v, nv, red);
  if (threadIdx.x == 0)
    for (int i = 0; i < nv; ++i) out[(size_t)blockIdx.x * nv + i] = v[i];
}
static inline int ksp_prered_rows(int nparts) { return (nparts + OX_PRERED_CHUNK - 1) / OX_PRERED_CHUNK; }

struct KspLayout {
  size_t state, state2, sums, partial, partial2, vec0, vec_stride, narrow0, narrow_stride, total;
  int nvec, nparts_max;
};

// `grid_max`: the largest number of partial-sum rows a mat-vec on the operator writes (0: the lane = row grid only).
// The LDS-window stream writes one row per WINDOW BLOCK, and with a short length-sort window (or few slices per block)
// there are more window blocks than lane = row blocks: the arrays are sized for whichever grid is larger.
static KspLayout ksp_layout(int64_t n_rows, int64_t n_cols, int ncomp, int ksp_type, int grid_max = 0) {
  KspLayout L;
  const int n_slices = (int)((n_rows + 63) / 64);
  const int nblk_spmv = (n_slices + 3) / 4;
  int nb8 = (nblk_spmv + 7) & ~7;
  if (grid_max > nb8) nb8 = (grid_max + 7) & ~7;
  L.nparts_max = nb8 + 16 > OX_VEC_MAX_BLOCKS ? nb8 + 16 : OX_VEC_MAX_BLOCKS;  // (+16: interior / boundary launches round up separately)
  L.nvec = (ksp_type == OX_KSP_CG || ksp_type == OX_KSP_CG_MERGED) ? 3 : (ksp_type == OX_KSP_CG_SINGLE ? 5 : 6);  // (both BiCGStab variants: 6)
  L.state = 0;
  L.state2 = ox_align(sizeof(KspState));  // (folded CG: the state ping-pongs between the two points of an iteration)
  L.sums = 2 * ox_align(sizeof(KspState));
  // sums, then the pre-reduction scratch of both partial arrays of a synchronisation point
  L.partial = L.sums + ox_align(sizeof(double) * (OX_PRERED_OFFSET + 2 * (size_t)(ksp_prered_rows(L.nparts_max) + 1) * OX_MAX_NV));
  L.partial2 = L.partial + ox_align(sizeof(double) * (size_t)L.nparts_max * 5 * OX_MAXC);
  L.vec0 = L.partial2 + ox_align(sizeof(double) * (size_t)L.nparts_max * 3 * OX_MAXC);
  L.vec_stride = ox_align(sizeof(double) * (size_t)n_cols * ncomp);
  L.narrow0 = L.vec0 + L.vec_stride * L.nvec;
  L.narrow_stride = ox_align(sizeof(double) * (size_t)n_cols);
  L.total = L.narrow0 + (ncomp > 1 ? L.narrow_stride * 8 : 0);  // compact vectors of the narrowed solve
  return L;
}

// the window grid of the operator, whatever slice list or column count a launch uses (every launch's grid is at most
// round8(n_wblocks), split launches of a partitioned operator at most 8 more: covered by the +16 above)
static inline int ksp_grid_max(const ox_sell *A) { return (A && A->wb_ptr && A->n_wblocks > 0) ? ((A->n_wblocks + 7) & ~7) + 8 : 0; }

extern "C" size_t ox_ksp_work_bytes(int64_t n_rows, int64_t n_cols, int ncomp, int ksp_type) {
  return ksp_layout(n_rows, n_cols, ncomp, ksp_type).total;
}
extern "C" size_t ox_ksp_work_bytes_for(const ox_sell *A, int ncomp, int ksp_type) {
  if (!A) return 0;
  return ksp_layout(A->n_rows, A->n_cols, ncomp, ksp_type, ksp_grid_max(A)).total;
}

// (per host thread: several ranks of a job may be driven from threads of ONE process, each inside its own solve --
// tests/test_gpu_threads_rehearsal.py; a thread's 1.2 KB of pinned memory and two events live as long as the process)
static thread_local KspState *g_state_host = nullptr;
// Where the last synchronisation point of the batch being queued writes the host's copy of the state (pinned, host
// mapped: g_state_host[1 + slot]); nullptr outside ksp_run_ahead.  An in-stream copy of these 400 bytes is a blit
// kernel of ~18 us plus ~6 us of stream bubble (rocprofv3, r03: 0.7 ms of a 16-ms pressure solve at 8 iterations a
// batch); the kernel's own posted stores cost it ~1 us.
static thread_local KspState *g_batch_mirror = nullptr;
static inline KspParams ksp_last_point(const KspParams &P, int k, int count) {
  KspParams Q = P;
  Q.mirror = (k + 1 == count) ? g_batch_mirror : nullptr;
  return Q;
}

template <int PH>
static int ksp_sync_point(KspState *S, double *partial, int nparts, int nv, double *sums,
                          const KspParams &P, const ox_dist *dist, hipStream_t st, KspPart2 B = KspPart2{nullptr, 0, 0}) {
  {
    double *scr = sums + OX_PRERED_OFFSET;
    // (the merged-reduction CG's scalar kernel requests up to 10 rows per thread in one round: worth a pre-reduction
    // launch from 64 K sums on only)
    const int64_t prered_min = PH == PH_CGM_IT ? 4 * OX_PRERED_MIN : OX_PRERED_MIN;
    if ((int64_t)nparts * nv >= prered_min) {
      const int g = ksp_prered_rows(nparts);
      hipLaunchKernelGGL(k_prereduce, dim3(g), dim3(256), 0, st, partial, nparts, nv, scr);
      partial = scr;
      nparts = g;
      scr += (size_t)g * nv;
    }
    if (B.nv > 0 && (int64_t)B.nparts * B.nv >= OX_PRERED_MIN) {
      const int g = ksp_prered_rows(B.nparts);
      hipLaunchKernelGGL(k_prereduce, dim3(g), dim3(256), 0, st, B.partial, B.nparts, B.nv, scr);
      B.partial = scr;
      B.nparts = g;
    }
  }
  const int nmax = nparts > B.nparts ? nparts : B.nparts;
  if (!dist) {
    hipLaunchKernelGGL((k_ksp_scalar<PH>), dim3(1), dim3(ox_red_threads(nmax)), 0, st, S, partial, nparts, nv, P, B);
    OX_LAUNCH_CHECK();
    return 0;
  }
  if (ox_prof_on) ox_prof_start(OX_TAG_SYNC_POINT, st, nv + B.nv);
  if (dist->p2p && dist->nranks <= 64) {  // (one rank: the self-loop plans of tools/predict_scaling.py)
    int threads = ox_red_threads(nmax);
    if (threads < ox_p2p_ar_threads(dist->nranks)) threads = ox_p2p_ar_threads(dist->nranks);
    hipLaunchKernelGGL((k_ksp_scalar_p2p<PH>), dim3(1), dim3(threads), 0, st, S, partial, nparts, nv, P,
                       ox_p2p_next_allreduce(dist), B);
    if (ox_prof_on) ox_prof_stop(st);
    OX_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((k_ksp_reduce<PH>), dim3(1), dim3(ox_red_threads(nmax)), 0, st, partial, nparts, nv, B, sums);
  OX_LAUNCH_CHECK();
  nv += B.nv;  // ONE all-reduce for both arrays
  if (ox_allreduce_impl(dist, sums, nv, st)) return -1;
  hipLaunchKernelGGL((k_ksp_logic<PH>), dim3(1), dim3(64), 0, st, S, sums, P);
  if (ox_prof_on) ox_prof_stop(st);
  OX_LAUNCH_CHECK();
  return 0;
}

// column extraction / insertion for the narrowed continuation
__global__ __launch_bounds__(256) void k_extract_col(int64_t n, const double *__restrict__ src, int nc,
                                                     int c, double *__restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i * nc + c];
}
__global__ __launch_bounds__(256) void k_insert_col(int64_t n, const double *__restrict__ src, int nc,
                                                    int c, double *__restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i * nc + c] = src[i];
}

struct KspVecs {
  double *x, *r, *z, *p, *q;       // CG (z, q) ...
  double *rhat, *v, *s, *t;        // ... BiCGStab
  double *u, *w;                   // single-reduction CG: u = D^-1 r, w = A u (s = A p by recurrence)
};

struct KspCtx {
  const ox_sell *A;
  const double *dinv;
  KspDinv D;  // dinv again, with its value dictionary where the caller has one (CG update kernels)
  KspState *S, *S2;
  double *sums, *partial, *partial2;
  const ox_dist *dist;
  hipStream_t st;
  int nb, nbs;
  int fold;  // blocks of the folded one-column CG update kernels (0: separate synchronisation points)
};

#define KSP_SYNC(PH, partial, nparts, nv)                                                       \
  do {                                                                                          \
    if (ksp_sync_point<PH>(C.S, partial, nparts, nv, C.sums, P, C.dist, C.st)) return -1;       \
  } while (0)

// blocks of the folded CG update kernels: the library's default (a per-solve value comes in through ox_ksp_options;
// the environment is read by the host layer, not here)
static int ksp_fold_blocks_default() {
  static int dflt = -1;
  if (dflt < 0) {
    // ONE block per compute unit: a CU that receives a second 1024-thread block takes twice as long as the others
    // (128^3 pressure: 256 blocks 59.1 us per iteration, 262 blocks 66.0, 512 blocks 65.8, 128 blocks 62.5)
    hipDeviceProp_t prop;
    int dev = 0;
    int ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    dflt = ncu > OX_FOLD_T ? OX_FOLD_T : ncu;  // (the second point reads one partial row per thread)
  }
  return dflt;
}
static inline int ksp_fold_blocks_of(int v) { return v < 0 ? ksp_fold_blocks_default() : (v > OX_FOLD_T ? OX_FOLD_T : v); }
template <int NC>
static int cg_iterations(const KspCtx &C, const KspVecs &V, const KspParams &P, int count) {
  const int64_t n = C.A->n_rows;
  const int *done = &C.S->done;
  if (NC == 1 && !C.dist && n >= 2 && C.fold > 0) {  // one column, one GPU: both points folded into the update kernels
    int64_t want = ((n >> 1) + OX_FOLD_T - 1) / OX_FOLD_T;
    if (want < 1) want = 1;
    const int nbf = (int)(want < C.fold ? want : C.fold);
    const int nbs1 = ox_spmv_dist_nparts(C.A, nullptr, 1);  // (block sums of the ONE-column mat-vec)
    for (int k = 0; k < count; ++k) {
      if (ox_spmv_dist(C.A, V.p, V.q, 1, OX_EPI_DOT, nullptr, nullptr, C.partial, done, nullptr, C.st)) return -1;
      const double *pin = C.partial;
      int npin = nbs1;
      if (npin >= OX_PRERED_MIN) {  // (a 17 M-row mass matrix: 66 K block sums)
        double *scr = C.sums + OX_PRERED_OFFSET;
        const int g = ksp_prered_rows(npin);
        hipLaunchKernelGGL(k_prereduce, dim3(g), dim3(256), 0, C.st, pin, npin, 1, scr);
        pin = scr;
        npin = g;
      }
      if (C.D.code) {
        hipLaunchKernelGGL(k_cg_update1f<true>, dim3(nbf), dim3(OX_FOLD_T), 0, C.st, n, C.S, C.S2, pin, npin, P, V.r, V.q,
                           C.D, C.partial2);
        hipLaunchKernelGGL(k_cg_update2f<true>, dim3(nbf), dim3(OX_FOLD_T), 0, C.st, n, C.S2, C.S, C.partial2, nbf,
                           ksp_last_point(P, k, count), V.x, V.r, C.D, V.p);
      } else {
        hipLaunchKernelGGL(k_cg_update1f<false>, dim3(nbf), dim3(OX_FOLD_T), 0, C.st, n, C.S, C.S2, pin, npin, P, V.r, V.q,
                           C.D, C.partial2);
        hipLaunchKernelGGL(k_cg_update2f<false>, dim3(nbf), dim3(OX_FOLD_T), 0, C.st, n, C.S2, C.S, C.partial2, nbf,
                           ksp_last_point(P, k, count), V.x, V.r, C.D, V.p);
      }
      OX_LAUNCH_CHECK();
    }
    return 0;
  }
  for (int k = 0; k < count; ++k) {
    if (ox_spmv_dist(C.A, V.p, V.q, NC, OX_EPI_DOT, nullptr, nullptr, C.partial, done, C.dist, C.st)) return -1;
    KSP_SYNC(PH_CG_A, C.partial, C.nbs, NC);
    hipLaunchKernelGGL((k_cg_update1<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.r, V.q, C.D, C.partial);
    OX_LAUNCH_CHECK();
    if (ksp_sync_point<PH_CG_B>(C.S, C.partial, C.nb, 2 * NC, C.sums, ksp_last_point(P, k, count), C.dist, C.st)) return -1;
    hipLaunchKernelGGL((k_cg_update2<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.x, V.r, C.D, V.p, 0);
    OX_LAUNCH_CHECK();
  }
  return 0;
}

// Merged-reduction CG: mat-vec with the three-sum epilogue, ONE synchronisation point (one all-reduce of
// {p.q, q.D^-1 q, r.z, z.z, p.q_old} in a partitioned run), one update kernel -- 3 kernels per iteration instead of 5.
// ---------------------------------------------------------------------------------------
// Merged-reduction CG with its ONE synchronisation point folded into the update kernel: 2 kernels per iteration
// (mat-vec with the two-sum epilogue, k_cgm_updatef).  Same construction as k_cg_update1f / 2f above: every block sums
// the mat-vec's partial sums {p.q, q.D^-1 q} and the previous update's {r.z, z.z, p.q_old}, thread 0 runs PH_CGM_IT on
// an LDS copy of the state, block 0 stores it -- into the OTHER state block: the state alternates S / S2 from one
// iteration to the next, so batches are even; the update's own partial sums alternate between two arrays likewise
// (a block reads the previous iteration's while others already write this one's).
// ---------------------------------------------------------------------------------------
#define OX_FOLDM_NP 2  // (the merged form is the default up to 2^20 rows: at most two pairs per thread)
template <bool CODE>
__global__ __launch_bounds__(OX_FOLD_T) void k_cgm_updatef(int64_t n, const KspState *Sin, KspState *Sout,
                                                           const double *__restrict__ pinA, int npinA,
                                                           const double *__restrict__ pinB, int npinB, KspParams P,
                                                           double *x, double *vr, const double *__restrict__ vq, KspDinv D,
                                                           double *vp, double *pout) {
  __shared__ double red[16 * 5], sums[5], dd[256];
  __shared__ KspState sh;
  const int T = OX_FOLD_T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * OX_FOLD_T, i0 = (int64_t)blockIdx.x * OX_FOLD_T + tid;
  // ---- the point: state, partial sums, dictionary requested first (unconditional, clamped), then the rows' operands
  const unsigned long long sw = reinterpret_cast<const unsigned long long *>(Sin)[min(tid, KSP_STATE_WORDS - 1)];
  double2 tA[10];
#pragma unroll
  for (int u = 0; u < 10; ++u) tA[u] = *reinterpret_cast<const double2 *>(pinA + 2 * (size_t)min(tid + u * T, npinA - 1));
  double tB[3];
  {
    const int pc = min(tid, max(npinB, 1) - 1);  // (npinB = 0 at the first point of a solve: the array is there, unread)
#pragma unroll
    for (int i = 0; i < 3; ++i) tB[i] = pinB[(size_t)pc * 3 + i];
  }
  double dv = 0.0;
  if constexpr (CODE) dv = D.dict[min(tid, D.n - 1)];
  double2 qq[OX_FOLDM_NP], rr[OX_FOLDM_NP], pp[OX_FOLDM_NP], xx[OX_FOLDM_NP], dq[OX_FOLDM_NP];
  unsigned short cq[OX_FOLDM_NP];
#pragma unroll
  for (int j = 0; j < OX_FOLDM_NP; ++j) {
    const int64_t iw = i0 + j * stride;
    const int64_t i = iw < n2 ? iw : min(i0, n2 - 1);
    qq[j] = *reinterpret_cast<const double2 *>(vq + 2 * i);
    rr[j] = *reinterpret_cast<const double2 *>(vr + 2 * i);
    pp[j] = *reinterpret_cast<const double2 *>(vp + 2 * i);
    xx[j] = *reinterpret_cast<const double2 *>(x + 2 * i);
    if constexpr (CODE) cq[j] = *reinterpret_cast<const unsigned short *>(D.code + 2 * i);
    else dq[j] = *reinterpret_cast<const double2 *>(D.v + 2 * i);
  }
  // (a fence the compiler cannot schedule across: every load above is issued before it -- "memory" --, and every use of
  // the point's operands depends on its outputs; left alone the selects and sums below were hoisted in front of the
  // rows' loads, which then waited behind the partial sums' round trip)
  int na = npinA, nb = npinB;
  asm volatile("" : "+s"(na), "+s"(nb) : : "memory");
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const bool in = tid + u * T < na;
    tA[u].x = in ? tA[u].x : 0.0;
    tA[u].y = in ? tA[u].y : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) tB[i] = tid < nb ? tB[i] : 0.0;
  reinterpret_cast<unsigned long long *>(&sh)[min(tid, KSP_STATE_WORDS - 1)] = sw;
  if constexpr (CODE) dd[min(tid, D.n - 1)] = dv;
  double v[5] = {0.0, 0.0, tB[0], tB[1], tB[2]};
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    v[0] += tA[u].x;
    v[1] += tA[u].y;
  }
  if (na > 10 * T) {
    for (int p = tid + 10 * T; p < na; p += T) {
      v[0] += pinA[2 * (size_t)p];
      v[1] += pinA[2 * (size_t)p + 1];
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double sv = ox_wave_sum(v[i]);
    if (lane == 0) red[i * 16 + wave] = sv;
  }
  ox_lds_barrier();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      double sv = lane < nw ? red[i * 16 + lane] : 0.0;
      sv += __shfl_down(sv, 8, 64);
      sv += __shfl_down(sv, 4, 64);
      sv += __shfl_down(sv, 2, 64);
      sv += __shfl_down(sv, 1, 64);
      v[i] = sv;
    }
    if (tid == 0 && !sh.done) {
#pragma unroll
      for (int i = 0; i < 5; ++i) sums[i] = v[i];
      for (int c = 0; c < P.nc; ++c) ksp_logic<PH_CGM_IT>(&sh, sums, c, P);
      ksp_finish(&sh, P.nc_total);
    }
  }
  ox_lds_barrier();
  if (blockIdx.x == 0) {
    ksp_state_store(Sout, &sh);
    if (P.mirror) ksp_state_store(P.mirror, &sh);
  }
  if (sh.done) return;  // (uniform)
  // ---- the rows: x += alpha p;  r -= alpha q;  z = D^-1 r;  p = z + beta p;  sums of r.z, z.z, p.q
  const double alpha = sh.alpha[P.c0], beta = sh.beta[P.c0];
  double s[3] = {0.0, 0.0, 0.0};
  auto rows = [&](int64_t e, const double2 q2, double2 r, double2 p, double2 xv, const double d0, const double d1) {
    xv.x = fma(alpha, p.x, xv.x);
    xv.y = fma(alpha, p.y, xv.y);
    r.x = fma(-alpha, q2.x, r.x);
    r.y = fma(-alpha, q2.y, r.y);
    const double z0 = d0 * r.x, z1 = d1 * r.y;
    p.x = fma(beta, p.x, z0);
    p.y = fma(beta, p.y, z1);
    *reinterpret_cast<double2 *>(x + e) = xv;
    *reinterpret_cast<double2 *>(vr + e) = r;
    *reinterpret_cast<double2 *>(vp + e) = p;
    s[0] = fma(r.x, z0, s[0]);
    s[1] = fma(z0, z0, s[1]);
    s[2] = fma(p.x, q2.x, s[2]);
    s[0] = fma(r.y, z1, s[0]);
    s[1] = fma(z1, z1, s[1]);
    s[2] = fma(p.y, q2.y, s[2]);
  };
#pragma unroll
  for (int j = 0; j < OX_FOLDM_NP; ++j) {
    const int64_t i = i0 + j * stride;
    if (i < n2) rows(2 * i, qq[j], rr[j], pp[j], xx[j], CODE ? dd[cq[j] & 0xff] : dq[j].x, CODE ? dd[cq[j] >> 8] : dq[j].y);
  }
  for (int64_t i = i0 + OX_FOLDM_NP * stride; i < n2; i += stride) {
    const int64_t e = 2 * i;
    rows(e, *reinterpret_cast<const double2 *>(vq + e), *reinterpret_cast<const double2 *>(vr + e),
         *reinterpret_cast<const double2 *>(vp + e), *reinterpret_cast<const double2 *>(x + e), KSP_DINV(e), KSP_DINV(e + 1));
  }
  if ((n & 1) && blockIdx.x == 0 && tid == 0) {
    const int64_t e = n - 1;
    const double pe = vp[e], qe = vq[e];
    x[e] = fma(alpha, pe, x[e]);
    const double r = fma(-alpha, qe, vr[e]), z = KSP_DINV(e) * r;
    const double pn = fma(beta, pe, z);
    vr[e] = r;
    vp[e] = pn;
    s[0] = fma(r, z, s[0]);
    s[1] = fma(z, z, s[1]);
    s[2] = fma(pn, qe, s[2]);
  }
  __syncthreads();  // (red is reused)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sv = ox_wave_sum(s[i]);
    if (lane == 0) red[i * 16 + wave] = sv;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double sv = lane < nw ? red[i * 16 + lane] : 0.0;
      sv += __shfl_down(sv, 8, 64);
      sv += __shfl_down(sv, 4, 64);
      sv += __shfl_down(sv, 2, 64);
      sv += __shfl_down(sv, 1, 64);
      if (lane == 0) pout[(size_t)blockIdx.x * 3 + i] = sv;
    }
  }
}

template <int NC>
static int cgm_iterations(const KspCtx &C, const KspVecs &V, const KspParams &P, int count, bool first) {
  const int64_t n = C.A->n_rows;
  const int *done = &C.S->done;
  if (NC == 1 && !C.dist && n >= 2 && C.fold > 0 && (count & 1) == 0) {
    // folded: the state alternates S / S2 per iteration (an even batch leaves it in S, where the host and the next
    // batch look for it), the update's partial sums between two arrays
    const int nbs1 = ox_spmv_dist_nparts(C.A, nullptr, 1);
    if (nbs1 <= 10 * OX_FOLD_T) {  // (every thread's <= 10 partial rows of the mat-vec are requested in one round)
      int64_t want = ((n >> 1) + OX_FOLD_T - 1) / OX_FOLD_T;
      if (want < 1) want = 1;
      const int nbf = (int)(want < C.fold ? want : C.fold);
      double *pu[2] = {C.partial2, C.partial2 + 3 * OX_FOLD_T};
      for (int k = 0; k < count; ++k) {
        KspState *Sin = (k & 1) ? C.S2 : C.S, *Sout = (k & 1) ? C.S : C.S2;
        ox_spmv_set_epilogue_dinv(C.D.code, C.D.dict);
        const int rc = ox_spmv_dist(C.A, V.p, V.q, 1, OX_EPI_CG_M2, C.dinv, nullptr, C.partial, &Sin->done, nullptr, C.st);
        ox_spmv_set_epilogue_dinv(nullptr, nullptr);
        if (rc) return -1;
        KspParams Q = ksp_last_point(P, k, count);
        Q.first = (first && k == 0) ? 1 : 0;
        const int npb = Q.first ? 0 : nbf;
        if (C.D.code)
          hipLaunchKernelGGL(k_cgm_updatef<true>, dim3(nbf), dim3(OX_FOLD_T), 0, C.st, n, Sin, Sout, C.partial, nbs1,
                             pu[(k + 1) & 1], npb, Q, V.x, V.r, V.q, C.D, V.p, pu[k & 1]);
        else
          hipLaunchKernelGGL(k_cgm_updatef<false>, dim3(nbf), dim3(OX_FOLD_T), 0, C.st, n, Sin, Sout, C.partial, nbs1,
                             pu[(k + 1) & 1], npb, Q, V.x, V.r, V.q, C.D, V.p, pu[k & 1]);
        OX_LAUNCH_CHECK();
      }
      return 0;
    }
  }
  for (int k = 0; k < count; ++k) {
    ox_spmv_set_epilogue_dinv(C.D.code, C.D.dict);
    const int rc = ox_spmv_dist(C.A, V.p, V.q, NC, OX_EPI_CG_M2, C.dinv, nullptr, C.partial, done, C.dist, C.st);
    ox_spmv_set_epilogue_dinv(nullptr, nullptr);
    if (rc) return -1;
    KspParams Q = ksp_last_point(P, k, count);
    Q.first = (first && k == 0) ? 1 : 0;
    if (ksp_sync_point<PH_CGM_IT>(C.S, C.partial, C.nbs, 2 * NC, C.sums, Q, C.dist, C.st,
                                  Q.first ? KspPart2{nullptr, 0, 0} : KspPart2{C.partial2, C.nb, 3 * NC}))
      return -1;
    hipLaunchKernelGGL((k_cgm_update<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.x, V.r, V.q, C.D, V.p, C.partial2);
    OX_LAUNCH_CHECK();
  }
  return 0;
}

// Single-reduction CG: update kernel, SpMV with the w.u epilogue, ONE synchronisation point (one
// all-reduce of {r.u, u.u, w.u} in a partitioned run) -- 3 kernels per iteration instead of 5.
template <int NC>
static int cgs_iterations(const KspCtx &C, const KspVecs &V, const KspParams &P, int count) {
  const int64_t n = C.A->n_rows;
  const int *done = &C.S->done;
  for (int k = 0; k < count; ++k) {
    hipLaunchKernelGGL((k_cgs_update<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.x, V.r, V.u, V.p, V.s, V.w,
                       C.dinv, C.partial);
    OX_LAUNCH_CHECK();
    if (ox_spmv_dist(C.A, V.u, V.w, NC, OX_EPI_DOT, nullptr, nullptr, C.partial2, done, C.dist, C.st)) return -1;
    if (ksp_sync_point<PH_CGS_IT>(C.S, C.partial, C.nb, 2 * NC, C.sums, ksp_last_point(P, k, count), C.dist, C.st,
                                  KspPart2{C.partial2, C.nbs, NC}))
      return -1;
  }
  return 0;
}

template <int NC>
static int bcgs_iterations(const KspCtx &C, const KspVecs &V, const KspParams &P, int count, bool first) {
  const int64_t n = C.A->n_rows;
  const int *done = &C.S->done;
  for (int k = 0; k < count; ++k) {
    if (!(first && k == 0)) {  // the first iteration's p = r was written by k_bcgs_init
      hipLaunchKernelGGL((k_bcgs_p<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.r, V.p, V.v, V.rhat);
      OX_LAUNCH_CHECK();
    }
    if (ox_spmv_dist(C.A, V.p, V.v, NC, OX_EPI_BCGS_V, C.dinv, V.rhat, C.partial, done, C.dist, C.st)) return -1;
    KSP_SYNC(PH_BCGS_1, C.partial, C.nbs, NC);
    hipLaunchKernelGGL((k_bcgs_s<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.r, V.v, V.s);
    OX_LAUNCH_CHECK();
    if (ox_spmv_dist(C.A, V.s, V.t, NC, OX_EPI_BCGS_T, C.dinv, nullptr, C.partial, done, C.dist, C.st)) return -1;
    KSP_SYNC(PH_BCGS_2, C.partial, C.nbs, 2 * NC);
    hipLaunchKernelGGL((k_bcgs_x<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.x, V.r, V.rhat, V.p, V.s,
                       V.t, C.partial, 0);
    OX_LAUNCH_CHECK();
    if (ksp_sync_point<PH_BCGS_3>(C.S, C.partial, C.nb, 2 * NC, C.sums, ksp_last_point(P, k, count), C.dist, C.st)) return -1;
  }
  return 0;
}

// Merged-reduction BiCGStab (OX_KSP_BCGS_MERGED): TWO synchronisation points per iteration instead of three --
// rhat.v behind the first mat-vec; {t.t, t.s, rhat.s, rhat.t, s.s} behind the second give omega, rho' and |r|
// together (PH_BCGSM_B) -- i.e. two all-reduces per iteration on a partitioned operator.
// The x / r update runs after the second point with that iteration's alpha and omega, fused with the next p
// update (k_bcgs_xp); the update the `done` flag skips at the end of the solve is applied by bcgsm_finish.
template <int NC>
static int bcgsm_iterations(const KspCtx &C, const KspVecs &V, const KspParams &P, int count, bool first) {
  const int64_t n = C.A->n_rows;
  const int *done = &C.S->done;
  (void)first;  // the first iteration's p = r was written by k_bcgs_init; every later p comes out of k_bcgs_xp
  for (int k = 0; k < count; ++k) {
    if (ox_spmv_dist(C.A, V.p, V.v, NC, OX_EPI_BCGS_V, C.dinv, V.rhat, C.partial, done, C.dist, C.st)) return -1;
    if (ksp_sync_point<PH_BCGS_1>(C.S, C.partial, C.nbs, NC, C.sums, P, C.dist, C.st)) return -1;
    hipLaunchKernelGGL((k_bcgs_s<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.r, V.v, V.s);
    OX_LAUNCH_CHECK();
    if (ox_spmv_dist(C.A, V.s, V.t, NC, OX_EPI_BCGS_T5, C.dinv, V.rhat, C.partial, done, C.dist, C.st)) return -1;
    if (ksp_sync_point<PH_BCGSM_B>(C.S, C.partial, C.nbs, 5 * NC, C.sums, ksp_last_point(P, k, count), C.dist, C.st)) return -1;
    hipLaunchKernelGGL((k_bcgs_xp<NC>), dim3(C.nb), dim3(256), 0, C.st, n, C.S, P.c0, V.x, V.r, V.rhat, V.p, V.s, V.t, V.v, 0);
    OX_LAUNCH_CHECK();
  }
  return 0;
}
template <int NC>
static int bcgsm_finish(const KspCtx &C, const KspVecs &V, const KspParams &P) {
  hipLaunchKernelGGL((k_bcgs_xp<NC>), dim3(C.nb), dim3(256), 0, C.st, C.A->n_rows, C.S, P.c0, V.x, V.r, V.rhat, V.p, V.s, V.t,
                     V.v, 1);
  OX_LAUNCH_CHECK();
  return 0;
}

static int ksp_read_state(const KspCtx &C) {
  OX_HIP(hipMemcpyAsync(g_state_host, C.S, sizeof(KspState), hipMemcpyDeviceToHost, C.st));
  OX_HIP(hipStreamSynchronize(C.st));
  return C.dist ? ox_dist_status(C.dist) : 0;  // a timed-out peer wait fails the solve
}

// One-column solves (the pressure CG, the narrowed tail of the velocity solves) run ONE BATCH AHEAD of
// the state the host looks at: batch k+1 is queued before the copy of the state after batch k is waited
// for, so the GPU never idles for the ~30 us of a read-back (the "host reads" of SURVEY section 7).  The
// price is one batch of no-op kernels after convergence, hence small batches.  g_state_host[1], [2] are
// the two copies in flight; the final state ends up in g_state_host[0].
static thread_local hipEvent_t g_state_ev[2] = {nullptr, nullptr};
template <class Iterate>
static int ksp_run_ahead(const KspCtx &C, Iterate &&iterate, int batch, int &it, int max_it) {
  for (int i = 0; i < 2; ++i)
    if (!g_state_ev[i]) OX_HIP(hipEventCreateWithFlags(&g_state_ev[i], hipEventDisableTiming));
  // batch -> state in g_state_host[1 + slot]: written by the batch's last synchronisation point itself (g_batch_mirror)
  auto queue = [&](int slot) -> int {
    g_state_host[1 + slot].done = 0;
    g_batch_mirror = g_state_host + 1 + slot;
    const int rc = iterate(batch);
    g_batch_mirror = nullptr;
    if (rc) return -1;
    OX_HIP(hipEventRecord(g_state_ev[slot], C.st));
    return 0;
  };
  if (queue(0)) return -1;
  it += batch;
  int cur = 0;
  for (;;) {
    const bool more = it <= max_it;  // (the kernels stop by themselves at max_it: reason DIVERGED_ITS)
    if (more) {
      if (queue(1 - cur)) return -1;
      it += batch;
    }
    OX_HIP(hipEventSynchronize(g_state_ev[cur]));
    if (C.dist && ox_dist_status(C.dist)) return -1;
    if (g_state_host[1 + cur].done || !more) {
      OX_HIP(hipStreamSynchronize(C.st));  // the batch queued ahead (no-ops after `done`) and its copy
      if (!more && !g_state_host[1 + cur].done) {  // ran out of batches: take the newest state
        OX_HIP(hipMemcpyAsync(g_state_host, C.S, sizeof(KspState), hipMemcpyDeviceToHost, C.st));
        OX_HIP(hipStreamSynchronize(C.st));
        return 0;
      }
      g_state_host[0] = g_state_host[1 + cur];
      return 0;
    }
    cur = 1 - cur;
  }
}

template <int NC>
static int ksp_run(int ksp_type, const ox_sell *A, const double *dinv, const double *b, double *x,
                   const KspParams &P, int guess, int check_every, char *work,
                   ox_ksp_result *result, const ox_dist *dist, hipStream_t st, const double *ax0, const KspDinv &Dc,
                   int fold_blocks, int run_ahead) {
  const int64_t n = A->n_rows;
  const KspLayout L = ksp_layout(A->n_rows, A->n_cols, NC, ksp_type, ksp_grid_max(A));
  KspCtx C;
  C.A = A;
  C.dinv = dinv;
  C.D = Dc;
  C.S = reinterpret_cast<KspState *>(work + L.state);
  C.S2 = reinterpret_cast<KspState *>(work + L.state2);
  C.sums = reinterpret_cast<double *>(work + L.sums);
  C.partial = reinterpret_cast<double *>(work + L.partial);
  C.partial2 = reinterpret_cast<double *>(work + L.partial2);
  C.dist = dist;
  C.st = st;
  C.nb = ox_vec_blocks(2 * (n > 0 ? n : 1));  // one row per thread
  C.nbs = ox_spmv_dist_nparts(A, dist, NC);
  C.fold = fold_blocks;
  double *vec[6];
  for (int i = 0; i < L.nvec; ++i) vec[i] = reinterpret_cast<double *>(work + L.vec0 + L.vec_stride * i);
  if (!g_state_host) OX_HIP(hipHostMalloc(&g_state_host, 3 * sizeof(KspState)));
  const bool cg = ksp_type == OX_KSP_CG, cgs = ksp_type == OX_KSP_CG_SINGLE, bm = ksp_type == OX_KSP_BCGS_MERGED;
  const bool cgm = ksp_type == OX_KSP_CG_MERGED;  // (one right-hand side only: ox_ksp_solve_dc)
  KspVecs V{};
  V.x = x;
  if (cgs) {
    V.r = vec[0], V.u = vec[1], V.p = vec[2], V.s = vec[3], V.w = vec[4];
    if (guess && !ax0) {
      if (ox_spmv_dist(A, x, V.w, NC, OX_EPI_NONE, nullptr, nullptr, nullptr, nullptr, dist, st)) return -1;
    }
    hipLaunchKernelGGL((k_cgs_init<NC>), dim3(C.nb), dim3(256), 0, st, n, b, x, (guess && ax0) ? ax0 : V.w, dinv, V.r, V.u,
                       V.p, V.s, guess, C.partial);
    OX_LAUNCH_CHECK();
    if (ox_spmv_dist(A, V.u, V.w, NC, OX_EPI_DOT, nullptr, nullptr, C.partial2, nullptr, dist, st)) return -1;
    if (ksp_sync_point<PH_CGS_INIT>(C.S, C.partial, C.nb, 3 * NC, C.sums, P, dist, st, KspPart2{C.partial2, C.nbs, NC}))
      return -1;
  } else if (cg || cgm) {
    V.r = vec[0], V.p = vec[1], V.q = vec[2];
    if (guess && !ax0) {
      if (ox_spmv_dist(A, x, V.q, NC, OX_EPI_NONE, nullptr, nullptr, nullptr, nullptr, dist, st)) return -1;
    }
    hipLaunchKernelGGL((k_cg_init<NC>), dim3(C.nb), dim3(256), 0, st, n, b, x, (guess && ax0) ? ax0 : V.q, dinv, V.r,
                       V.p, guess, C.partial);
    OX_LAUNCH_CHECK();
    KSP_SYNC(PH_CG_INIT, C.partial, C.nb, 3 * NC);
  } else {
    V.r = vec[0], V.rhat = vec[1], V.p = vec[2], V.v = vec[3], V.s = vec[4], V.t = vec[5];
    if (guess && !ax0) {
      if (ox_spmv_dist(A, x, V.t, NC, OX_EPI_NONE, nullptr, nullptr, nullptr, nullptr, dist, st)) return -1;
    }
    hipLaunchKernelGGL((k_bcgs_init<NC>), dim3(C.nb), dim3(256), 0, st, n, b, x, (guess && ax0) ? ax0 : V.t, dinv, V.r,
                       V.rhat, V.p, V.v, guess, C.partial);
    OX_LAUNCH_CHECK();
    KSP_SYNC(PH_BCGS_INIT, C.partial, C.nb, 2 * NC);
  }
  bool bm_first = true;  // BiCGStab: the first iteration of the solve finds p = r in place (k_bcgs_init)
  auto iterate = [&](auto nc_tag, const KspVecs &W, const KspParams &Q) -> int {
    constexpr int N_ = decltype(nc_tag)::value;
    const bool first = bm_first;
    bm_first = false;
    return cgs ? cgs_iterations<N_>(C, W, Q, check_every)
           : cgm ? cgm_iterations<N_>(C, W, Q, check_every, first)
               : (cg ? cg_iterations<N_>(C, W, Q, check_every)
                     : (bm ? bcgsm_iterations<N_>(C, W, Q, check_every, first) : bcgs_iterations<N_>(C, W, Q, check_every, first)));
  };
  // small batches (the tail of no-ops after convergence is about one batch), large enough for the host
  // to stay ahead: 8 pressure-CG iterations are 40 launches for 600 us of GPU time
  auto batch_of = [&](int every) { return every > 8 ? 8 : every; };
  int it = 0;  // iterations queued so far (the kernels stop by themselves at max_it)
  bool tail_done = false;  // the narrowed continuation applied the deferred updates of its column itself
  // The merged-reduction BiCGStab re-enters this loop when the STORED residual of a column fails the test its
  // recurrence norm passed (PH_BCGSM_FIN); every other method leaves it after one pass.
  for (;;) {
    tail_done = false;
    if (NC == 1 && run_ahead) {
      const int bsz = batch_of(check_every);
      auto it1 = [&](int count) -> int {
        const bool first = bm_first;
        bm_first = false;
        return cgs ? cgs_iterations<1>(C, V, P, count)
               : cgm ? cgm_iterations<1>(C, V, P, count, first)
                   : (cg ? cg_iterations<1>(C, V, P, count)
                         : (bm ? bcgsm_iterations<1>(C, V, P, count, first) : bcgs_iterations<1>(C, V, P, count, first)));
      };
      if (ksp_run_ahead(C, it1, bsz, it, P.max_it)) return -1;
    }
    for (; !(NC == 1 && run_ahead) && it <= P.max_it; it += check_every) {
      if (iterate(std::integral_constant<int, NC>{}, V, P)) return -1;
      if (ksp_read_state(C)) return -1;
      if (g_state_host->done) break;
      if (NC > 1) {
        // Narrowing: the columns run in lockstep, so once all but one have converged the rest of
        // the solve would drag NC-wide vectors along for nothing.  Extract the live column into
        // compact vectors and continue with the 1-column kernels (same recurrences, same scalars:
        // the state block is addressed through P.c0); insert x back at the end.
        int live = -1, nlive = 0;
        for (int c = 0; c < NC; ++c)
          if (g_state_host->active[c]) live = c, ++nlive;
        if (nlive == 1) {
          double *cv[8];
          for (int i = 0; i < 8; ++i) cv[i] = reinterpret_cast<double *>(work + L.narrow0 + L.narrow_stride * i);
          KspVecs W{};
          double *src[8] = {V.x, V.r, V.z, V.p, V.q, V.rhat, V.v, nullptr};
          double **dst[8] = {&W.x, &W.r, &W.z, &W.p, &W.q, &W.rhat, &W.v, nullptr};
          if (cg) src[2] = nullptr;  // no z vector: the second update kernel forms D^-1 r itself
          if (cgs) {  // x, r, u, p, s (a recurrence: carried over), w
            src[2] = V.u, dst[2] = &W.u;
            src[4] = V.s, dst[4] = &W.s;
            src[5] = V.w, dst[5] = &W.w;
            src[6] = nullptr;
          }
          for (int i = 0; i < 7; ++i) {
            *dst[i] = cv[i];
            if (!src[i]) continue;
            hipLaunchKernelGGL(k_extract_col, dim3(C.nb), dim3(256), 0, st, n, src[i], NC, live, cv[i]);
            OX_LAUNCH_CHECK();
          }
          if (!cgs) {
            W.s = cv[2];  // BiCGStab temporaries share the slots CG uses for z and q
            W.t = cv[4];
          }
          KspParams P1 = P;
          P1.nc = 1;
          P1.c0 = live;
          if (run_ahead) {
            const int bsz = batch_of(check_every);
            auto it1 = [&](int count) -> int {
              return cgs ? cgs_iterations<1>(C, W, P1, count)
                         : (cg ? cg_iterations<1>(C, W, P1, count)
                               : (bm ? bcgsm_iterations<1>(C, W, P1, count, false) : bcgs_iterations<1>(C, W, P1, count, false)));
            };
            it += check_every;
            if (ksp_run_ahead(C, it1, bsz, it, P.max_it)) return -1;
          } else {
            for (it += check_every; it <= P.max_it; it += check_every) {
              if (iterate(std::integral_constant<int, 1>{}, W, P1)) return -1;
              if (ksp_read_state(C)) return -1;
              if (g_state_host->done) break;
            }
          }
          if (cg) {  // the last x += alpha p (see k_cg_update2)
            hipLaunchKernelGGL((k_cg_update2<1>), dim3(C.nb), dim3(256), 0, st, n, C.S, P1.c0, W.x, W.r, C.D, W.p, 1);
            OX_LAUNCH_CHECK();
          }
          if (bm && g_state_host->its[live] > 0) {
            if (bcgsm_finish<1>(C, W, P1)) return -1;
            // (the re-test below reads the stored residuals of all columns from the NC-wide block)
            hipLaunchKernelGGL(k_insert_col, dim3(C.nb), dim3(256), 0, st, n, W.r, NC, live, V.r);
            OX_LAUNCH_CHECK();
          }
          tail_done = true;
          hipLaunchKernelGGL(k_insert_col, dim3(C.nb), dim3(256), 0, st, n, W.x, NC, live, x);
          OX_LAUNCH_CHECK();
          break;
        }
      }
    }
    if (!g_state_host->done) OX_FAIL("ox_ksp_solve: device state never reported completion");
    if (!bm) break;
    bool any = false;
    for (int c = 0; c < NC; ++c) any = any || g_state_host->its[c] > 0;
    if (!any) break;  // every column converged on the initial (stored) residual: nothing ran, nothing to re-test
    // the x / r update of the last iteration (skipped by `done`)
    if (!tail_done && bcgsm_finish<NC>(C, V, P)) return -1;
    bool retest = false;
    for (int c = 0; c < NC; ++c)
      retest = retest || g_state_host->reason[c] == OX_CONVERGED_RTOL || g_state_host->reason[c] == OX_CONVERGED_ATOL;
    if (!retest) break;
    hipLaunchKernelGGL((k_rr<NC>), dim3(C.nb), dim3(256), 0, st, n, V.r, C.partial);
    OX_LAUNCH_CHECK();
    KSP_SYNC(PH_BCGSM_FIN, C.partial, C.nb, NC);
    if (ksp_read_state(C)) return -1;
    if (g_state_host->done) break;  // every stored residual passes (or a column ran out of iterations)
    // resume: rhat <- r, p <- r in the re-opened columns (k_bcgs_p reads the restart flags); p = r in the others
    hipLaunchKernelGGL((k_bcgs_p<NC>), dim3(C.nb), dim3(256), 0, st, n, C.S, P.c0, V.r, V.p, V.v, V.rhat);
    OX_LAUNCH_CHECK();
    it = 0;
    for (int c = 0; c < NC; ++c) it = g_state_host->its[c] > it ? g_state_host->its[c] : it;
  }
  if (cg && !tail_done) {  // the last x += alpha p (see k_cg_update2)
    hipLaunchKernelGGL((k_cg_update2<NC>), dim3(C.nb), dim3(256), 0, st, n, C.S, P.c0, V.x, V.r, C.D, V.p, 1);
    OX_LAUNCH_CHECK();
  }
  for (int c = 0; c < NC; ++c) {
    result->reason[c] = g_state_host->reason[c];
    result->its[c] = g_state_host->its[c];
    result->rnorm[c] = g_state_host->rn[c];
    result->bnorm[c] = g_state_host->bn[c];
    result->resumed[c] = bm ? g_state_host->nresume[c] : 0;
  }
  return 0;
}

extern "C" int ox_ksp_options_default(ox_ksp_options *o) {
  if (!o) OX_FAIL("ox_ksp_options_default: null argument");
  memset(o, 0, sizeof(*o));
  o->rtol = 1e-5;    // PETSc's KSP defaults (KSPCreate; -ksp_view: "relative=1e-05, absolute=1e-50, divergence=10000.")
  o->atol = 1e-50;
  o->divtol = 1e4;
  o->max_it = 10000;
  o->check_every = 1;
  o->fold_blocks = -1;
  o->run_ahead = -1;
  return 0;
}

extern "C" int ox_ksp_default_fold_blocks(void) { return ksp_fold_blocks_default(); }

extern "C" int ox_ksp_kernels_per_iteration(int ksp_type, const ox_sell *A, int ncomp, int check_every, int fold_blocks,
                                            int partitioned) {
  // what cg_iterations / cgm_iterations / ... launch per iteration for this operator (reporting: bench.py)
  if (!A) return -1;
  if (ksp_type == OX_KSP_CG_MERGED && ncomp > 1) ksp_type = OX_KSP_CG;
  const bool one = ncomp == 1 && !partitioned && A->n_rows >= 2 && ksp_fold_blocks_of(fold_blocks) > 0;
  const int batch = check_every > 8 ? 8 : (check_every < 1 ? 1 : check_every);
  switch (ksp_type) {
    case OX_KSP_CG: {
      if (!one) return 5;
      return ox_spmv_dist_nparts(A, nullptr, 1) >= OX_PRERED_MIN ? 4 : 3;  // (+ the pre-reduction of many block sums)
    }
    case OX_KSP_CG_MERGED:
      return (one && (batch & 1) == 0 && ox_spmv_dist_nparts(A, nullptr, 1) <= 10 * OX_FOLD_T) ? 2 : 3;
    case OX_KSP_CG_SINGLE: return 3;
    case OX_KSP_BCGS: return 8;
    case OX_KSP_BCGS_MERGED: return 6;
    default: return -1;
  }
}

extern "C" int ox_ksp_solve_opt(int ksp_type, const ox_sell *A, const double *dinv, const double *b, double *x, int ncomp,
                                const ox_ksp_options *opt, void *work, size_t work_bytes, ox_ksp_result *result,
                                const ox_dist *dist, void *stream) {
  if (!A || !dinv || !b || !x || !work || !result || !opt) OX_FAIL("ox_ksp_solve: null argument");
  if (ksp_type != OX_KSP_CG && ksp_type != OX_KSP_BCGS && ksp_type != OX_KSP_CG_SINGLE && ksp_type != OX_KSP_BCGS_MERGED &&
      ksp_type != OX_KSP_CG_MERGED)
    OX_FAIL("ox_ksp_solve: ksp_type=%d", ksp_type);
  if (ksp_type == OX_KSP_CG_MERGED && ncomp > 1) ksp_type = OX_KSP_CG;  // (the merged form is the one-column solver)
  if (ncomp < 1 || ncomp > OX_MAXC) OX_FAIL("ox_ksp_solve: ncomp=%d out of range", ncomp);
  if (work_bytes < ox_ksp_work_bytes_for(A, ncomp, ksp_type))
    OX_FAIL("ox_ksp_solve: workspace too small (%zu < %zu)", work_bytes, ox_ksp_work_bytes_for(A, ncomp, ksp_type));
  if (opt->dinv_code && (!opt->dinv_dict || opt->n_dinv_dict < 1 || opt->n_dinv_dict > 256))
    OX_FAIL("ox_ksp_solve: dinv dictionary of %d entries", opt->n_dinv_dict);
  if (!(opt->divtol > 0.0)) OX_FAIL("ox_ksp_solve: divtol=%g (must be positive; PETSc's default is 1e4)", opt->divtol);
  const int max_it = opt->max_it < 1 ? 1 : opt->max_it;
  const int check_every = opt->check_every < 1 ? 1 : opt->check_every;
  memset(result, 0, sizeof(*result));
  KspParams P{};
  P.rtol = opt->rtol;
  P.atol = opt->atol;
  P.dtol = opt->divtol;
  P.max_it = max_it;
  P.nc = ncomp;
  P.c0 = 0;
  P.nc_total = ncomp;
  P.max_restarts = opt->max_restarts < 0 ? 0 : opt->max_restarts;
  hipStream_t st = ox_stream(stream);
  char *w = static_cast<char *>(work);
  const KspDinv D{dinv, opt->dinv_code, opt->dinv_dict, opt->dinv_code ? opt->n_dinv_dict : 0};
  const int fold = ksp_fold_blocks_of(opt->fold_blocks), ahead = opt->run_ahead < 0 ? 1 : (opt->run_ahead ? 1 : 0);
  const int guess = opt->nonzero_guess;
  const double *ax0 = opt->ax0;
  switch (ncomp) {
    case 1: return ksp_run<1>(ksp_type, A, dinv, b, x, P, guess, check_every, w, result, dist, st, ax0, D, fold, ahead);
    case 2: return ksp_run<2>(ksp_type, A, dinv, b, x, P, guess, check_every, w, result, dist, st, ax0, D, fold, ahead);
    default: return ksp_run<3>(ksp_type, A, dinv, b, x, P, guess, check_every, w, result, dist, st, ax0, D, fold, ahead);
  }
}

extern "C" int ox_ksp_solve_dc(int ksp_type, const ox_sell *A, const double *dinv, const double *b,
                               double *x, int ncomp, double rtol, double atol, int max_it,
                               int nonzero_guess, int check_every, int max_restarts, void *work,
                               size_t work_bytes, ox_ksp_result *result, const ox_dist *dist, void *stream,
                               const double *ax0, const uint8_t *dinv_code, const double *dinv_dict, int n_dinv_dict) {
  ox_ksp_options o;
  ox_ksp_options_default(&o);
  o.rtol = rtol;
  o.atol = atol;
  o.max_it = max_it;
  o.nonzero_guess = nonzero_guess;
  o.check_every = check_every;
  o.max_restarts = max_restarts;
  o.ax0 = ax0;
  o.dinv_code = dinv_code;
  o.dinv_dict = dinv_dict;
  o.n_dinv_dict = n_dinv_dict;
  return ox_ksp_solve_opt(ksp_type, A, dinv, b, x, ncomp, &o, work, work_bytes, result, dist, stream);
}

extern "C" int ox_ksp_solve_ax0(int ksp_type, const ox_sell *A, const double *dinv, const double *b,
                                double *x, int ncomp, double rtol, double atol, int max_it,
                                int nonzero_guess, int check_every, int max_restarts, void *work,
                                size_t work_bytes, ox_ksp_result *result, const ox_dist *dist, void *stream,
                                const double *ax0) {
  return ox_ksp_solve_dc(ksp_type, A, dinv, b, x, ncomp, rtol, atol, max_it, nonzero_guess, check_every, max_restarts,
                         work, work_bytes, result, dist, stream, ax0, nullptr, nullptr, 0);
}

extern "C" int ox_ksp_solve(int ksp_type, const ox_sell *A, const double *dinv, const double *b,
                            double *x, int ncomp, double rtol, double atol, int max_it,
                            int nonzero_guess, int check_every, int max_restarts, void *work,
                            size_t work_bytes, ox_ksp_result *result, const ox_dist *dist, void *stream) {
  return ox_ksp_solve_dc(ksp_type, A, dinv, b, x, ncomp, rtol, atol, max_it, nonzero_guess, check_every, max_restarts,
                         work, work_bytes, result, dist, stream, nullptr, nullptr, nullptr, 0);
}
