// Device-side state and scalar logic of the Krylov loops (ox_ksp.hip), shared with the SpMV kernels
// (ox_spmv.hip) whose fused epilogues END a synchronisation point themselves: the block that arrives last
// reduces the per-block partial sums (fixed order) and runs the scalar recurrences -- no separate
// one-block kernel between the producer and the next vector kernel ("folded" synchronisation point).
#pragma once
#include "ox_kernels.h"

struct KspState {
  double rz[OX_MAXC], alpha[OX_MAXC], beta[OX_MAXC], omega[OX_MAXC], rho[OX_MAXC];
  double bn[OX_MAXC], rn[OX_MAXC];
  int active[OX_MAXC], reason[OX_MAXC], its[OX_MAXC];
  int restart[OX_MAXC];   // BiCGStab: next k_bcgs_p re-seeds rhat <- r for this column
  int nrestart[OX_MAXC];
  int done;  // all components finished
  int pad[2];
};

struct KspParams {
  double rtol, atol;
  int max_it;
  int nc;        // columns of THIS launch (sums are indexed 0..nc-1)
  int c0;        // state column of launch column 0 (narrowed continuation: the one live column)
  int nc_total;  // columns of the solve
  int max_restarts;  // BiCGStab restarts allowed on a rho/omega breakdown (0 = PETSc: report -5)
};

enum { PH_CG_INIT = 0, PH_CG_A, PH_CG_B, PH_BCGS_INIT, PH_BCGS_1, PH_BCGS_2, PH_BCGS_3, PH_CGS_INIT, PH_CGS_IT, PH_BCGSM_B, PH_COUNT };

__device__ __forceinline__ int ksp_test(double rn, double bn, const KspParams &P) {
  if (!(rn == rn) || isinf(rn)) return OX_DIVERGED_NANORINF;
  if (rn <= P.atol) return OX_CONVERGED_ATOL;
  if (rn <= P.rtol * bn) return OX_CONVERGED_RTOL;
  return 0;
}

// Scalar logic of one synchronisation point, run by thread c for component c.
// `s` holds the globally reduced sums of that point.
template <int PH>
__device__ __forceinline__ void ksp_logic(KspState *S, const double *s_all, int cl, const KspParams &P) {
  const int NC = P.nc;
  const int c = P.c0 + cl;          // state column
  const double *s = s_all + cl - c;  // so that s[c], s[NC + c], ... address launch column cl
  if (PH == PH_CG_INIT) {  // s = {r.z, z.z, (Db).(Db)}
    S->rz[c] = s[c];
    S->rn[c] = sqrt(s[NC + c]);
    S->bn[c] = sqrt(s[2 * NC + c]);
    S->its[c] = 0;
    S->alpha[c] = 0.0;
    S->beta[c] = 0.0;
    const int r = ksp_test(S->rn[c], S->bn[c], P);
    S->reason[c] = r;
    S->active[c] = (r == 0);
  } else if (PH == PH_CG_A) {  // s = {p.q}
    if (S->active[c]) {
      const double pq = s[c];
      if (pq == 0.0 || !(pq == pq)) {
        S->reason[c] = (pq == pq) ? OX_DIVERGED_BREAKDOWN : OX_DIVERGED_NANORINF;
        S->active[c] = 0;
        S->alpha[c] = 0.0;
      } else {
        S->alpha[c] = S->rz[c] / pq;
      }
    } else {
      S->alpha[c] = 0.0;
    }
  } else if (PH == PH_CG_B) {  // s = {r.z (new), z.z}
    if (S->active[c]) {
      S->its[c] += 1;
      S->rn[c] = sqrt(s[NC + c]);
      int r = ksp_test(S->rn[c], S->bn[c], P);
      if (r == 0 && S->its[c] >= P.max_it) r = OX_DIVERGED_ITS;
      if (r) {
        S->reason[c] = r;
        S->active[c] = 0;
        S->beta[c] = 0.0;
      } else {
        S->beta[c] = s[c] / S->rz[c];
        S->rz[c] = s[c];
      }
    } else {
      S->beta[c] = 0.0;
    }
  } else if (PH == PH_CGS_INIT) {  // s = {r.u, u.u, (Db).(Db), w.u}   (u = D^-1 r, w = A u)
    S->rz[c] = s[c];
    S->rn[c] = sqrt(s[NC + c]);
    S->bn[c] = sqrt(s[2 * NC + c]);
    S->its[c] = 0;
    S->beta[c] = 0.0;
    int r = ksp_test(S->rn[c], S->bn[c], P);
    const double wu = s[3 * NC + c];
    if (r == 0 && (wu == 0.0 || !(wu == wu))) r = (wu == wu) ? OX_DIVERGED_BREAKDOWN : OX_DIVERGED_NANORINF;
    S->reason[c] = r;
    S->active[c] = (r == 0);
    S->alpha[c] = r == 0 ? s[c] / wu : 0.0;
  } else if (PH == PH_CGS_IT) {  // s = {r.u, u.u (new residual), w.u}: the ONE reduction of the iteration
    if (S->active[c]) {
      S->its[c] += 1;
      S->rn[c] = sqrt(s[NC + c]);
      int r = ksp_test(S->rn[c], S->bn[c], P);
      if (r == 0 && S->its[c] >= P.max_it) r = OX_DIVERGED_ITS;
      if (r == 0) {
        // beta = gamma'/gamma;  alpha = gamma' / (delta - beta gamma'/alpha)   (Chronopoulos & Gear)
        const double g = s[c], beta = g / S->rz[c];
        const double den = s[2 * NC + c] - beta * g / S->alpha[c];
        if (den == 0.0 || !(den == den)) r = (den == den) ? OX_DIVERGED_BREAKDOWN : OX_DIVERGED_NANORINF;
        else {
          S->beta[c] = beta;
          S->alpha[c] = g / den;
          S->rz[c] = g;
        }
      }
      if (r) {
        S->reason[c] = r;
        S->active[c] = 0;
        S->alpha[c] = 0.0;
        S->beta[c] = 0.0;
      }
    } else {
      S->alpha[c] = 0.0;
      S->beta[c] = 0.0;
    }
  } else if (PH == PH_BCGS_INIT) {  // s = {r.r, (Db).(Db)}  (r already preconditioned)
    S->rn[c] = sqrt(s[c]);
    S->bn[c] = sqrt(s[NC + c]);
    S->its[c] = 0;
    S->restart[c] = 0;
    S->nrestart[c] = 0;
    const int r = ksp_test(S->rn[c], S->bn[c], P);
    S->reason[c] = r;
    S->active[c] = (r == 0);
    // (p = v = 0 in the first k_bcgs_p: any omega serves; 0 for a column that is finished already, so that the
    // merged variant's deferred x update adds nothing to it)
    S->alpha[c] = S->omega[c] = (r == 0) ? 1.0 : 0.0;
    // first iteration: rho_new = rhat.r = r.r, beta = (rho_new/1)*(1/1); p = v = 0
    S->rho[c] = s[c];
    S->beta[c] = S->active[c] ? s[c] : 0.0;
    if (S->active[c] && s[c] == 0.0) {
      S->reason[c] = OX_DIVERGED_BREAKDOWN;
      S->active[c] = 0;
    }
  } else if (PH == PH_BCGS_1) {  // s = {rhat.v}
    S->restart[c] = 0;  // consumed by the k_bcgs_p that ran just before
    if (S->active[c]) {
      const double rv = s[c];
      if (rv == 0.0 || !(rv == rv)) {
        S->reason[c] = (rv == rv) ? OX_DIVERGED_BREAKDOWN : OX_DIVERGED_NANORINF;
        S->active[c] = 0;
        S->alpha[c] = 0.0;
        S->omega[c] = 0.0;  // (merged variant: its deferred x update must add nothing for this column)
      } else {
        S->alpha[c] = S->rho[c] / rv;
      }
    } else {
      S->alpha[c] = 0.0;
      S->omega[c] = 0.0;
    }
  } else if (PH == PH_BCGS_2) {  // s = {t.t, t.s}
    if (S->active[c]) {
      const double tt = s[c];
      S->omega[c] = (tt != 0.0) ? s[NC + c] / tt : 0.0;
    } else {
      S->omega[c] = 0.0;
    }
  } else if (PH == PH_BCGSM_B) {
    // Merged-reduction BiCGStab, the ONE reduction behind t = D^-1 A s:  s = {t.t, t.s, rhat.s, rhat.t, s.s}
    //   omega = t.s / t.t;   r = s - omega t  =>  |r|^2 = s.s - 2 omega t.s + omega^2 t.t,   rho' = rhat.s - omega rhat.t
    // (what PH_BCGS_2 and PH_BCGS_3 take from two reductions; the residual norm by recurrence, as in every
    // merged / pipelined BiCGStab).  The x update with this iteration's alpha and omega runs AFTER this point.
    if (S->active[c]) {
      const double tt = s[c], ts = s[NC + c], hs = s[2 * NC + c], ht = s[3 * NC + c], ss = s[4 * NC + c];
      const double om = (tt != 0.0) ? ts / tt : 0.0;
      S->omega[c] = om;
      S->its[c] += 1;
      double rr = fma(om, fma(om, tt, -2.0 * ts), ss);
      if (rr < 0.0) rr = 0.0;  // round-off of the recurrence near convergence
      S->rn[c] = sqrt(rr);
      const double rho_new = fma(-om, ht, hs);
      int r = ksp_test(S->rn[c], S->bn[c], P);
      bool reseed = false;
      if (r == 0 && (om == 0.0 || rho_new == 0.0)) {
        if (S->nrestart[c] < P.max_restarts && rr > 0.0) reseed = true;
        else r = OX_DIVERGED_BREAKDOWN;
      }
      if (r == 0 && S->its[c] >= P.max_it) r = OX_DIVERGED_ITS;
      if (r) {
        S->reason[c] = r;
        S->active[c] = 0;
        S->beta[c] = 0.0;
      } else if (reseed) {
        S->restart[c] = 1;
        S->nrestart[c] += 1;
        S->rho[c] = rr;  // rhat.r with rhat = r
        S->beta[c] = 0.0;
      } else {
        S->beta[c] = (rho_new / S->rho[c]) * (S->alpha[c] / om);
        S->rho[c] = rho_new;
      }
    }
    // (an inactive column keeps alpha = omega = 0 from the first point of the iteration: x is left alone)
  } else if (PH == PH_BCGS_3) {  // s = {r.r, rhat.r}
    if (S->active[c]) {
      S->its[c] += 1;
      S->rn[c] = sqrt(s[c]);
      int r = ksp_test(S->rn[c], S->bn[c], P);
      bool reseed = false;
      if (r == 0 && (S->omega[c] == 0.0 || s[NC + c] == 0.0)) {
        // rho = rhat.r = 0 (or omega = 0) with an unconverged residual: PETSc stops here.  With
        // restarts allowed, re-seed the shadow residual (rhat <- r, p <- r) and carry on.
        if (S->nrestart[c] < P.max_restarts && s[c] > 0.0) reseed = true;
        else r = OX_DIVERGED_BREAKDOWN;
      }
      if (r == 0 && S->its[c] >= P.max_it) r = OX_DIVERGED_ITS;
      if (r) {
        S->reason[c] = r;
        S->active[c] = 0;
        S->beta[c] = 0.0;
      } else if (reseed) {
        S->restart[c] = 1;
        S->nrestart[c] += 1;
        S->rho[c] = s[c];  // rhat.r with rhat = r
        S->beta[c] = 0.0;  // p = r
        S->alpha[c] = 1.0;
        S->omega[c] = 1.0;
      } else {
        S->beta[c] = (s[NC + c] / S->rho[c]) * (S->alpha[c] / S->omega[c]);
        S->rho[c] = s[NC + c];
      }
    } else {
      S->beta[c] = 0.0;
    }
  }
}

__device__ __forceinline__ void ksp_finish(KspState *S, int nc) {
  int any = 0;
  for (int c = 0; c < nc; ++c) any |= S->active[c];
  S->done = !any;
}

// The state block is staged in LDS: one coalesced load, the (dependent, branchy) scalar logic at
// LDS latency, one coalesced store -- instead of ~10 serialized global round trips per phase.
static_assert(sizeof(KspState) % 8 == 0, "KspState is copied in 8-byte words");
constexpr int KSP_STATE_WORDS = sizeof(KspState) / 8;

__device__ __forceinline__ void ksp_state_load(KspState *sh, const KspState *S) {
  if (threadIdx.x < KSP_STATE_WORDS)
    reinterpret_cast<unsigned long long *>(sh)[threadIdx.x] =
        reinterpret_cast<const unsigned long long *>(S)[threadIdx.x];
}
__device__ __forceinline__ void ksp_state_store(KspState *S, const KspState *sh) {
  if (threadIdx.x < KSP_STATE_WORDS)
    reinterpret_cast<unsigned long long *>(S)[threadIdx.x] =
        reinterpret_cast<const unsigned long long *>(sh)[threadIdx.x];
}


constexpr bool ksp_is_init(int ph) { return ph == PH_CG_INIT || ph == PH_BCGS_INIT || ph == PH_CGS_INIT; }

// ---------------------------------------------------------------------------------------------
// Folded synchronisation point: the kernel that produces the dot products also ends the synchronisation
// point, so no one-block kernel (6 us + a kernel boundary) sits between it and the next vector kernel.
//
//   * every WORK block stores its block sums as tagged granules -- each double as two 8-byte words
//     {epoch : 32 | half of the value : 32}, written with agent-scope relaxed atomic stores (sc1, write-through)
//     -- and ends.  Nothing to wait for: no fence, no ticket, no returning atomic (a first version drew tickets
//     from group counters: the two dependent fabric round trips at the end of every block cost the 7-us
//     blocks of the pressure SpMV 12 us per launch, more than the kernel it replaced);
//   * OX_FOLD_R extra REDUCER blocks at the end of the grid sweep the granules of their share of the rows with
//     sc1 loads until every tag carries this launch's epoch (the data is the flag: the guide's R2 granule,
//     MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16), sum them in a fixed order and publish the
//     group sum the same way; the last reducer block sweeps the OX_FOLD_R group sums, runs the scalar
//     recurrences of the phase on the state block and stores it for the next kernel.
// Work blocks never wait for anything, so the reducers cannot deadlock them; the reducers' sweeps are bounded
// (OX_FOLD_SPINS): a time-out ends the solve with OX_DIVERGED_FOLD_TIMEOUT instead of hanging the GPU.
// The order of every sum is fixed and is mirrored by the unfolded path (k_prereduce with the same OX_FOLD_R
// row ranges, then k_ksp_scalar): bit-identical scalars either way (tests/test_gpu_krylov_variants.py).
// Epochs count the folded launches of one solve (the granules are zeroed when the solve starts).
// ---------------------------------------------------------------------------------------------
#define OX_FOLD_R 8
#define OX_FOLD_MAX_ROWS 16384  // producers with more blocks keep the separate kernels (their kernels run for ~1 ms)
#define OX_FOLD_SPINS 400000
#define OX_DIVERGED_FOLD_TIMEOUT (-98)

typedef unsigned long long ox_u64;

struct KspFoldArgs {  // per (phase, narrowing) entry of a small device table: read by the last reducer only
  KspState *S;
  KspParams P;
  int phase;
};

struct KspFold {      // kernel argument (kept small: it lives in SGPRs for the whole kernel)
  const KspFoldArgs *args;  // nullptr: not folded -- the blocks store plain partial sums
  ox_u64 *gran;             // [rows][nv][2] tagged granules, then [OX_FOLD_R][nv][2] group sums at ggran
  ox_u64 *ggran;
  unsigned epoch;
  int nwb;                  // work blocks of this launch (the grid has OX_FOLD_R more)
};

__device__ __forceinline__ void ksp_logic_rt(int ph, KspState *S, const double *s, int cl, const KspParams &P) {
  switch (ph) {
    case PH_CG_INIT: ksp_logic<PH_CG_INIT>(S, s, cl, P); break;
    case PH_CG_A: ksp_logic<PH_CG_A>(S, s, cl, P); break;
    case PH_CG_B: ksp_logic<PH_CG_B>(S, s, cl, P); break;
    case PH_BCGS_INIT: ksp_logic<PH_BCGS_INIT>(S, s, cl, P); break;
    case PH_BCGS_1: ksp_logic<PH_BCGS_1>(S, s, cl, P); break;
    case PH_BCGS_2: ksp_logic<PH_BCGS_2>(S, s, cl, P); break;
    case PH_BCGS_3: ksp_logic<PH_BCGS_3>(S, s, cl, P); break;
    case PH_BCGSM_B: ksp_logic<PH_BCGSM_B>(S, s, cl, P); break;
    default: break;
  }
}

__device__ __forceinline__ void ox_gran_store(ox_u64 *g, unsigned epoch, double v) {
  const ox_u64 b = (ox_u64)__double_as_longlong(v), e = (ox_u64)epoch << 32;
  __hip_atomic_store(g, e | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, e | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns whether both halves carry the epoch
__device__ __forceinline__ bool ox_gran_load(const ox_u64 *g, unsigned epoch, double &v) {
  const ox_u64 lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const ox_u64 hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
  return (unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch;
}

// LDS a producer kernel sets aside for its block sum and for the reducers of a folded synchronisation point
struct KspFoldLds {
  double red[16 * OX_MAX_NV];
  KspState st;
  int flag;
};

// rows [r0, r1) of group j when nparts rows are split over OX_FOLD_R reducers (the unfolded path uses the same)
__host__ __device__ __forceinline__ int ox_fold_rpg(int nparts) { return (nparts + OX_FOLD_R - 1) / OX_FOLD_R; }

// Sweep `cnt` tagged rows of nv values (thread t: rows t, t + T, ... summed in that order, a few rows in flight)
// until every tag matches; then the ordered block sum.  Returns false on a time-out (uniform).
template <int NV>
__device__ __forceinline__ bool ksp_fold_sweep(const ox_u64 *gran, int cnt, unsigned epoch, double (&v)[OX_MAX_NV],
                                               KspFoldLds &L) {
  const int T = blockDim.x;
  for (int spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < OX_MAX_NV; ++i) v[i] = 0.0;
    constexpr int U = NV >= 8 ? 1 : 8 / NV;  // rows in flight per thread (registers: the producer's own loop sets the budget)
    for (int p0 = threadIdx.x; p0 < cnt; p0 += U * T) {
      double t[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * T;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          t[u][i] = 0.0;
          if (p < cnt) ok = ox_gran_load(gran + ((size_t)p * NV + i) * 2, epoch, t[u][i]) && ok;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (p0 + u * T < cnt) v[i] += t[u][i];
    }
    if (__syncthreads_and(ok ? 1 : 0)) break;
    if (spins >= OX_FOLD_SPINS) return false;
    __builtin_amdgcn_s_sleep(4);
  }
  ox_block_sum_wide(v, NV, L.red);
  return true;
}

// All threads of a 256-thread block call this at the end of a producer kernel.  Work blocks (blockIdx.x <
// F.nwb): s holds the block sums in thread 0 (after ox_block_sum_256).  Reducer blocks: s is ignored.
template <int NV>
__device__ __forceinline__ void ksp_arrive(const double (&s)[NV], double *__restrict__ partial, const KspFold &F,
                                           KspFoldLds &L) {
  const int b = blockIdx.x;
  if (!F.args) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) partial[(size_t)b * NV + i] = s[i];
    }
    return;
  }
  if (b < F.nwb) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) ox_gran_store(F.gran + ((size_t)b * NV + i) * 2, F.epoch, s[i]);
    }
    return;
  }
  // ---- reducer j: its share of the rows
  const int j = b - F.nwb, rpg = ox_fold_rpg(F.nwb);
  const int r0 = min(F.nwb, j * rpg), r1 = min(F.nwb, r0 + rpg);
  double v[OX_MAX_NV];
  bool good = ksp_fold_sweep<NV>(F.gran + (size_t)r0 * NV * 2, r1 - r0, F.epoch, v, L);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i)  // a timed-out group publishes NaN: the last reducer then sees it at once
      ox_gran_store(F.ggran + ((size_t)j * NV + i) * 2, F.epoch, good ? v[i] : __longlong_as_double(0x7ff8000000000000ll));
  }
  if (j != OX_FOLD_R - 1) return;
  // ---- last reducer: the group sums (k_ksp_scalar's arithmetic over OX_FOLD_R rows), then the scalar logic
  const KspFoldArgs A = *F.args;
  ksp_state_load(&L.st, A.S);
  __syncthreads();  // L.red is free again
  good = ksp_fold_sweep<NV>(F.ggran, OX_FOLD_R, F.epoch, v, L) && good;
  if (threadIdx.x == 0) {
    if (!good) {  // a reducer gave up waiting (a work block never stored): end the solve loudly
      for (int c = 0; c < A.P.nc_total; ++c) {
        if (L.st.active[c] || ksp_is_init(A.phase)) L.st.reason[c] = OX_DIVERGED_FOLD_TIMEOUT;
        L.st.active[c] = 0;
      }
      L.st.done = 1;
    } else if (ksp_is_init(A.phase) || !L.st.done) {
      for (int c = 0; c < A.P.nc; ++c) ksp_logic_rt(A.phase, &L.st, v, c, A.P);
      ksp_finish(&L.st, A.P.nc_total);
    }
  }
  __syncthreads();
  ksp_state_store(A.S, &L.st);
}
