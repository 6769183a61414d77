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
// blockDim.x >= 64 >= nranks; stage: LDS [64][OX_P2P_MAXV + 1].  Result in vals after the call.
__device__ __forceinline__ void ox_p2p_allreduce_block(double *vals, int n, const ox_p2p_ar &a,
                                                       double (*stage)[OX_P2P_MAXV + 1]) {
  const int r = threadIdx.x;
  if (r < a.nranks) {
    char *dst = a.r_slot[r * 2 + a.parity];
    for (int i = 0; i < n; ++i)
      __hip_atomic_store(reinterpret_cast<double *>(dst) + i, vals[i], __ATOMIC_RELAXED, OX_SYS);
    // (ONE release: the fence; a release STORE behind it would write back and invalidate the L2 a second time)
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + OX_P2P_SLOT - 8), a.seq, __ATOMIC_RELAXED, OX_SYS);
    const char *src = a.my_slots + ((size_t)a.parity * a.nranks + r) * OX_P2P_SLOT;
    ox_p2p_wait(reinterpret_cast<const unsigned long long *>(src + OX_P2P_SLOT - 8), a.seq, a.timeout_ticks, a.err);
    for (int i = 0; i < n; ++i)
      stage[r][i] = __hip_atomic_load(reinterpret_cast<const double *>(src) + i, __ATOMIC_RELAXED, OX_SYS);
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    double s = 0.0;
    for (int q = 0; q < a.nranks; ++q) s += stage[q][threadIdx.x];
    vals[threadIdx.x] = s;
  }
  __syncthreads();
}

// host: the next all-reduce of a plan (advances its sequence counter)
ox_p2p_ar ox_p2p_next_allreduce(const ox_dist *d);
