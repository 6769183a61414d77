// Device-side state and scalar logic of the Krylov loops (ox_ksp.hip).
// (Round 3 tried to let the producer kernels end their synchronisation points themselves -- tagged write-through
// partial sums swept by reducer blocks of the same launch -- and took it out again: profiles/
// r03_krylov_sync_fold_experiment.txt.)
#pragma once
#include "ox_kernels.h"

struct KspState {
  double rz[OX_MAXC], alpha[OX_MAXC], beta[OX_MAXC], omega[OX_MAXC], rho[OX_MAXC];
  double bn[OX_MAXC], rn[OX_MAXC];
  int active[OX_MAXC], reason[OX_MAXC], its[OX_MAXC];
  int restart[OX_MAXC];   // BiCGStab: next k_bcgs_p re-seeds rhat <- r for this column
  int nrestart[OX_MAXC];
  int nresume[OX_MAXC];   // merged BiCGStab: times PH_BCGSM_FIN re-opened the column (stored residual above the tolerance)
  int done;  // all components finished
  int pad[3];
};

struct KspParams {
  double rtol, atol;
  double dtol;   // PETSc's -ksp_divtol: |r| >= dtol |b| ends the solve with OX_DIVERGED_DTOL (KSPConvergedDefault; default 1e4)
  int max_it;
  int nc;        // columns of THIS launch (sums are indexed 0..nc-1)
  int c0;        // state column of launch column 0 (narrowed continuation: the one live column)
  int nc_total;  // columns of the solve
  int max_restarts;  // BiCGStab restarts allowed on a rho/omega breakdown (0 = PETSc: report -5)
  int first;         // merged-reduction CG: this synchronisation point is the first of the solve (no update kernel ran yet)
  KspState *mirror;  // host-mapped copy of the state, written by the LAST synchronisation point of a batch (nullptr: none)
};

enum { PH_CG_INIT = 0, PH_CG_A, PH_CG_B, PH_BCGS_INIT, PH_BCGS_1, PH_BCGS_2, PH_BCGS_3, PH_CGS_INIT, PH_CGS_IT, PH_BCGSM_B, PH_CGM_IT, PH_BCGSM_FIN, PH_COUNT };

__device__ __forceinline__ int ksp_test(double rn, double bn, const KspParams &P) {
  if (!(rn == rn) || isinf(rn)) return OX_DIVERGED_NANORINF;
  if (rn <= P.atol) return OX_CONVERGED_ATOL;
  if (rn <= P.rtol * bn) return OX_CONVERGED_RTOL;
  // KSPConvergedDefault: rnorm >= divtol * rnorm0 with rnorm0 = the norm the relative test uses (|D^-1 b| here, for a
  // zero and for a nonzero initial guess alike).  (b = 0 with a nonzero guess makes PETSc report DTOL for ANY residual;
  // that quirk is not reproduced: the test needs |b| > 0.)
  if (bn > 0.0 && rn >= P.dtol * bn) return OX_DIVERGED_DTOL;
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
  } else if (PH == PH_CGM_IT) {
    // Merged-reduction CG, the ONE point of an iteration, behind q = A p:
    //   s = {p.q, q.D^-1 q | r.z, z.z, p.q_old of the update kernel that ran before (absent at the first point)}
    // with z = D^-1 r.  The update r' = r - alpha q gives  r'.z' = r.z - 2 alpha q.z + alpha^2 q.D^-1 q  by algebra, and
    // q.z = p.q - beta_old (q.p_old) because z = p - beta_old p_old, with q.p_old = p.(A p_old) = p.q_old (A symmetric),
    // which the update kernel summed when it formed p.  No orthogonality is assumed anywhere, so beta is known HERE
    // exactly as the two-point form would compute it, up to rounding, and x, r, p are updated by one kernel.  The
    // carried r.z is replaced by the true one at every point: no drift.  The residual norm of iteration k is the
    // true |z| its update kernel summed -- tested one point later (one surplus mat-vec per solve; the queued update
    // is a no-op).
    if (S->active[c]) {
      int r = 0;
      double pqo = 0.0;
      if (!P.first) {
        S->its[c] += 1;
        S->rz[c] = s[2 * NC + c];
        S->rn[c] = sqrt(s[3 * NC + c]);
        pqo = s[4 * NC + c];
        r = ksp_test(S->rn[c], S->bn[c], P);
        if (r == 0 && S->its[c] >= P.max_it) r = OX_DIVERGED_ITS;
      }
      const double pq = s[c];
      if (r == 0 && (pq == 0.0 || !(pq == pq))) r = (pq == pq) ? OX_DIVERGED_BREAKDOWN : OX_DIVERGED_NANORINF;
      if (r) {
        S->reason[c] = r;
        S->active[c] = 0;
        S->alpha[c] = 0.0;
        S->beta[c] = 0.0;
      } else {
        const double a = S->rz[c] / pq;
        const double qz = fma(-S->beta[c], pqo, pq);  // (beta of the previous iteration; 0 at the first point: p = z)
        const double rz_new = fma(a, fma(a, s[NC + c], -2.0 * qz), S->rz[c]);
        S->alpha[c] = a;
        S->beta[c] = rz_new / S->rz[c];
        S->rz[c] = rz_new;  // (replaced by the update kernel's own sum at the next point)
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
    S->nresume[c] = 0;
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
  } else if (PH == PH_BCGSM_FIN) {
    // Merged-reduction BiCGStab, once per solve, after the deferred x / r update: s = {r.r} of the STORED residual.
    // A column the recurrence norm of PH_BCGSM_B declared converged is tested again on it -- the test PETSc's KSPBCGS
    // makes every iteration (reference ksp.py:76) -- and carries on from a re-seeded shadow residual if it fails: near
    // 1e-10..1e-12 relative the recurrence s.s - 2 omega t.s + omega^2 t.t cancels and can sit below the true norm.
    const int rs = S->reason[c];
    if ((rs == OX_CONVERGED_RTOL || rs == OX_CONVERGED_ATOL) && S->its[c] > 0) {
      const double rt = sqrt(s[c]);
      S->rn[c] = rt;
      int r = ksp_test(rt, S->bn[c], P);
      if (r == 0 && S->its[c] >= P.max_it) r = OX_DIVERGED_ITS;
      if (r) {
        S->reason[c] = r;
      } else {  // resume: rhat <- r, p <- r by the k_bcgs_p the host queues next (not counted as a breakdown restart)
        S->reason[c] = 0;
        S->active[c] = 1;
        S->restart[c] = 1;
        S->nresume[c] += 1;
        S->rho[c] = s[c];
        S->beta[c] = 0.0;
        S->alpha[c] = 1.0;
        S->omega[c] = 1.0;
      }
    }
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


// (points that run whatever the `done` flag says: the initial ones, and the re-test of a finished merged BiCGStab)
constexpr bool ksp_is_init(int ph) { return ph == PH_CG_INIT || ph == PH_BCGS_INIT || ph == PH_CGS_INIT || ph == PH_BCGSM_FIN; }

// sums reduced at the synchronisation point of a phase, per right-hand side (sizes the register arrays of
// k_ksp_scalar: a 1024-thread block has 128 registers per thread, OX_MAX_NV-wide arrays spilled)
__host__ __device__ constexpr int ksp_ph_nv(int ph) {
  return ph == PH_CG_INIT ? 3 : ph == PH_CG_A ? 1 : ph == PH_CG_B ? 2 : ph == PH_BCGS_INIT ? 2 : ph == PH_BCGS_1 ? 1
       : ph == PH_BCGS_2 ? 2 : ph == PH_BCGS_3 ? 2 : ph == PH_CGS_INIT ? 4 : ph == PH_CGS_IT ? 3 : ph == PH_BCGSM_FIN ? 1 : 5;  // (PH_BCGSM_B, PH_CGM_IT: 5)
}
