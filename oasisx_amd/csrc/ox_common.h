// Shared device/host helpers of the oasisx MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/oasisx_hip.h"

#define OX_WAVE 64
#define OX_MAXC 3  // velocity components solved in lockstep

extern thread_local char ox_err_buf[512];

#define OX_FAIL(...)                                       \
  do {                                                     \
    snprintf(ox_err_buf, sizeof(ox_err_buf), __VA_ARGS__); \
    return -1;                                             \
  } while (0)

#define OX_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) OX_FAIL("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

#define OX_LAUNCH_CHECK() OX_HIP(hipGetLastError())

// Direct xGMI transport of a halo plan (ox_dist.hip): every rank owns one uncached window that
// its peers map through HIP IPC and write into (halo values, all-reduce contributions, sequence
// flags).  Host-side sequence counters advance with every enqueued exchange.
struct ox_p2p {
  char *win;             // this rank's window (device, uncached)
  size_t win_bytes;
  int64_t n_ghost;       // ghost dofs of this rank (staging = 2 x n_ghost x OX_MAXC doubles)
  int32_t *peers_dev;    // device [n_peers]
  int64_t *send_off_dev; // device [n_peers+1]
  double **r_stage;      // device [n_peers*2]: peer's staging base for parity 0/1
  int64_t *r_off;        // device [n_peers]: dof offset of this rank's block in the peer's ghosts
  unsigned long long **r_hflag;  // device [n_peers*2]: this rank's halo flag in the peer's window
  char **r_slot;         // device [nranks*2]: this rank's all-reduce slot in rank r's window
  unsigned *ticket;      // device: last-block detection of the push kernel
  int *err_dev;          // device: sticky time-out flag, written by the kernels (ox_dist_status)
  unsigned long long hseq, aseq;  // exchanges enqueued so far
  long long timeout_ticks;        // wall_clock64 ticks a kernel waits for a peer
  void **opened;         // host [n_opened]: IPC mappings to close
  int n_opened;
  // release protocol of the push kernel and the all-reduce (ox_dist_set_p2p_release): 1 = CONSERVATIVE (default): every
  // storing wave fences at system scope behind its payload stores and the flags are release stores; 0 = the round-5
  // fast form (s_waitcnt vmcnt(0) per wave, one system fence by the last block / per wave, relaxed flag stores), which
  // has only ever run with all ranks on ONE device
  int conservative;
};

// Halo plan + RCCL communicator (ox_dist.hip).  NULL everywhere = single GPU.
struct ox_dist {
  void *comm;  // ncclComm_t
  int rank, nranks, n_peers;
  int32_t *peers;        // host [n_peers]
  int64_t *send_off;     // host [n_peers+1] offsets into send_idx
  int64_t *recv_off;     // host [n_peers+1] offsets into the ghost block
  const int32_t *send_idx;  // device [send_off[n_peers]] owned rows to pack
  int64_t n_owned, n_ghost;
  double *send_buf;      // device [send_off[n_peers] * OX_MAXC]
  // user transport (ox_dist_create_custom): replaces the RCCL calls, same pack kernel / call sites
  int (*halo_cb)(void *user, const double *send_dev, double *ghost_dev, int ncomp);
  int (*allreduce_cb)(void *user, double *buf_dev, int n);
  void *user;
  ox_p2p *p2p;  // non-NULL: halo exchange and all-reduce go over the xGMI windows
  // RCCL transport, overlapped exchange: the pack + send/recv run on a side stream between two events
  hipStream_t side;
  hipEvent_t ev_begin, ev_done;
  int overlap;  // ox_dist_set_overlap: -1 the transport's default, 0 / 1 exchange-then-multiply / overlapped mat-vecs
};

// blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2): give every XCD a
// contiguous range of the work so neighbouring rows -- and their x gathers -- share an L2.
// Bijective for any nblk (speed only, never correctness).
__device__ __forceinline__ int ox_xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ double ox_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// Sum NV per-thread values over a 256-thread block (4 waves); result valid in thread 0.
template <int NV>
__device__ __forceinline__ void ox_block_sum_256(double (&v)[NV], double *lds /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = ox_wave_sum(v[i]);
    if (lane == 0) lds[wave * NV + i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = lds[i] + lds[NV + i] + lds[2 * NV + i] + lds[3 * NV + i];
  }
}

static inline hipStream_t ox_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Internal launchers shared between translation units.
int ox_halo_forward_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st);
// the same exchange in two halves: work enqueued on `st` between begin and end overlaps with it and
// must not touch the ghost block of x
int ox_halo_begin_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st);
int ox_halo_end_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st);
int ox_allreduce_impl(const ox_dist *d, double *buf, int n, hipStream_t st);
