// Device side of the direct xGMI transport (see ox_dist.hip): bounded waits on sequence flags
// and the in-block all-reduce over the ranks' windows, shared by the stand-alone all-reduce kernel
// and the fused Krylov scalar kernel.
#pragma once
#include "ox_common.h"

#define OX_P2P_SLOT 128  // bytes of one all-reduce slot: 15 doubles + the sequence flag
#define OX_P2P_MAXV 15
#define OX_SYS __HIP_MEMORY_SCOPE_SYSTEM

// everything one all-reduce needs on the device (passed by value to the kernels)
struct ox_p2p_ar {
  char *const *r_slot;   // device [nranks*2]: this rank's slot in rank r's window, per parity
  const char *my_slots;  // this rank's window: slots [2][nranks]
  int nranks, parity;
  unsigned long long seq;
  long long timeout_ticks;
  int *err;
  int conservative;  // release-scope flag stores (ox_dist_set_p2p_release)
};

// wait until *flag >= seq; false on time-out (and the sticky error is raised)
__device__ __forceinline__ bool ox_p2p_wait(const unsigned long long *flag, unsigned long long seq,
                                            long long timeout_ticks, int *err) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, OX_SYS)) return false;  // a peer is gone: do not wait again
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, OX_SYS) < seq) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, OX_SYS);
      return false;
    }
  }
  return true;
}

// Sum vals[0..n) over the ranks, in rank order (the same bits on every rank).  vals lives in LDS
// and is complete on entry (a __syncthreads() precedes the call); all threads of the block call;
// blockDim.x is a multiple of 64, ideally >= 16 * nranks (ox_p2p_ar_threads); stage: LDS [64][OX_P2P_MAXV + 1].
// Result in vals after the call.
// SIXTEEN lanes per rank: lane i of the group stores value i into that rank's slot, lane 15 raises the flag (word 15 of
// the slot) behind the wave's one release, waits for the rank's flag in its own window, and the sixteen lanes read the
// rank's values.  (Rounds 1-4: ONE thread per rank stored, and read back, its n values one after the other -- every
// system-scope access to the uncached window a round trip of its own: 2.7 us per value, 39 us for the fifteen sums of a
// three-column merged BiCGStab point.  Found by tracing the self-loop plans, round 5.)
__device__ __forceinline__ void ox_p2p_allreduce_block(double *vals, int n, const ox_p2p_ar &a,
                                                       double (*stage)[OX_P2P_MAXV + 1]) {
  const int T = blockDim.x;
  const int items = (a.nranks * 16 + 63) & ~63;  // whole waves take part in every trip
  for (int it = threadIdx.x; it < items; it += T) {
    const int r = it >> 4, i = it & 15;
    const bool live = r < a.nranks;
    char *dst = a.r_slot[(live ? r : 0) * 2 + a.parity];
    if (live && i < n) __hip_atomic_store(reinterpret_cast<double *>(dst) + i, vals[i], __ATOMIC_RELAXED, OX_SYS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's payload stores are acknowledged ...
    __threadfence_system();                            // ... ONE fence for all of them, then the flags (relaxed, or -- conservative plans -- release stores)
    const char *src = a.my_slots + ((size_t)a.parity * a.nranks + (live ? r : 0)) * OX_P2P_SLOT;
    if (live && i == 15) {
      // conservative plans (the default until the windows have crossed a link): the flag itself is a system-scope
      // RELEASE store behind the wave's fence
      if (a.conservative)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + OX_P2P_SLOT - 8), a.seq, __ATOMIC_RELEASE, OX_SYS);
      else
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + OX_P2P_SLOT - 8), a.seq, __ATOMIC_RELAXED, OX_SYS);
      ox_p2p_wait(reinterpret_cast<const unsigned long long *>(src + OX_P2P_SLOT - 8), a.seq, a.timeout_ticks, a.err);
    }
    // the wave reconverges behind the waits of its flag lanes: no load of a rank's values may be issued, or moved by
    // the compiler, in front of them
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (live && i < n) stage[r][i] = __hip_atomic_load(reinterpret_cast<const double *>(src) + i, __ATOMIC_RELAXED, OX_SYS);
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    double s = 0.0;
    for (int q = 0; q < a.nranks; ++q) s += stage[q][threadIdx.x];
    vals[threadIdx.x] = s;
  }
  __syncthreads();
}
// threads a block needs to give every rank its sixteen lanes in one trip
static inline int ox_p2p_ar_threads(int nranks) { return ((nranks * 16 + 63) / 64) * 64; }

// host: the next all-reduce of a plan (advances its sequence counter)
ox_p2p_ar ox_p2p_next_allreduce(const ox_dist *d);
