// Mesh-partitioned runs: one process per GPU over xGMI.  The halo exchange (DOLFINx
// scatter_forward, reference fracstep.py:453,497,502,551,632,655; ksp.py:77) and the all-reduce
// of the Krylov scalars have two transports behind the same call sites:
//  * RCCL: a pack kernel + one grouped ncclSend/ncclRecv per neighbour straight into the ghost
//    block of the vector; one small ncclAllReduce per synchronisation point;
//  * direct xGMI stores ("p2p"): every rank owns an uncached window that its peers map through HIP
//    IPC.  The push kernel gathers the interface values and stores them straight into the
//    neighbours' windows, followed by a sequence flag; the receiver's copy kernel waits for the
//    flags and moves the staged values into the ghost block.  The all-reduce is one single-block
//    kernel: store the contribution into every rank's window, wait for everybody's, sum in rank
//    order (bit-identical on all ranks).  No library call and ~2 small kernels per exchange: the
//    exchanges of a strong-scaled Krylov iteration are latency-, not bandwidth-bound.
//    Every wait is bounded (time-out -> sticky error flag -> the host call fails).
#include <rccl/rccl.h>

#include <stdlib.h>

#include "ox_kernels.h"
#include "ox_p2p.h"

#define OX_NCCL(call)                                                                       \
  do {                                                                                      \
    ncclResult_t r_ = (call);                                                               \
    if (r_ != ncclSuccess) OX_FAIL("%s:%d %s: %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
  } while (0)

extern "C" int ox_comm_unique_id(char *id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  OX_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int ox_comm_create(const char *id128, int rank, int nranks, void **comm_out) {
  if (!id128 || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) OX_FAIL("ox_comm_create: bad argument");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm;
  OX_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
  *comm_out = comm;
  return 0;
}

extern "C" int ox_comm_info(void *comm, int *nranks, int *rank, int *device) {
  if (!comm) OX_FAIL("ox_comm_info: null communicator");
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  int v = 0;
  if (nranks) { OX_NCCL(ncclCommCount(c, &v)); *nranks = v; }
  if (rank) { OX_NCCL(ncclCommUserRank(c, &v)); *rank = v; }
  if (device) { OX_NCCL(ncclCommCuDevice(c, &v)); *device = v; }
  return 0;
}

extern "C" int ox_comm_destroy(void *comm) {
  if (comm) ncclCommDestroy(static_cast<ncclComm_t>(comm));
  return 0;
}

extern "C" int ox_dist_create(void *comm, int rank, int nranks, int n_peers, const int32_t *peers,
                              const int64_t *send_off, const int32_t *send_idx_dev,
                              const int64_t *recv_off, int64_t n_owned, int64_t n_ghost,
                              ox_dist **out) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) OX_FAIL("ox_dist_create: bad argument");
  ox_dist *d = static_cast<ox_dist *>(calloc(1, sizeof(ox_dist)));
  if (!d) OX_FAIL("ox_dist_create: out of memory");
  d->comm = comm;
  d->overlap = -1;
  d->rank = rank;
  d->nranks = nranks;
  d->n_peers = n_peers;
  d->peers = static_cast<int32_t *>(malloc(sizeof(int32_t) * (n_peers + 1)));
  d->send_off = static_cast<int64_t *>(malloc(sizeof(int64_t) * (n_peers + 1)));
  d->recv_off = static_cast<int64_t *>(malloc(sizeof(int64_t) * (n_peers + 1)));
  d->send_off[0] = d->recv_off[0] = 0;
  for (int p = 0; p < n_peers; ++p) d->peers[p] = peers[p];
  for (int p = 0; p <= n_peers && n_peers > 0; ++p) {
    d->send_off[p] = send_off[p];
    d->recv_off[p] = recv_off[p];
  }
  d->send_idx = send_idx_dev;
  d->n_owned = n_owned;
  d->n_ghost = n_ghost;
  if (n_peers > 0 && d->recv_off[n_peers] != n_ghost) OX_FAIL("ox_dist_create: recv_off/n_ghost mismatch");
  const int64_t ns = n_peers > 0 ? d->send_off[n_peers] : 0;
  if (ns > 0) OX_HIP(hipMalloc(&d->send_buf, sizeof(double) * ns * OX_MAXC));
  *out = d;
  return 0;
}

extern "C" int ox_dist_create_custom(int rank, int nranks, int n_peers, const int32_t *peers,
                                     const int64_t *send_off, const int32_t *send_idx_dev,
                                     const int64_t *recv_off, int64_t n_owned, int64_t n_ghost,
                                     int (*halo_cb)(void *, const double *, double *, int),
                                     int (*allreduce_cb)(void *, double *, int), void *user,
                                     ox_dist **out) {
  if (!halo_cb || !allreduce_cb) OX_FAIL("ox_dist_create_custom: null callback");
  static int dummy_comm;
  int rc = ox_dist_create(&dummy_comm, rank, nranks, n_peers, peers, send_off, send_idx_dev, recv_off,
                          n_owned, n_ghost, out);
  if (rc) return rc;
  (*out)->comm = nullptr;
  (*out)->halo_cb = halo_cb;
  (*out)->allreduce_cb = allreduce_cb;
  (*out)->user = user;
  return 0;
}

extern "C" int ox_memcpy(void *dst, const void *src, size_t bytes, int to_device, void *stream) {
  hipStream_t st = ox_stream(stream);
  OX_HIP(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// direct xGMI transport
// ---------------------------------------------------------------------------------------------
struct P2pLayout {
  size_t ar, hf, st, st_stride, total;
};
static P2pLayout p2p_layout(int nranks, int64_t n_ghost) {
  P2pLayout L;
  L.ar = 0;
  L.hf = (size_t)2 * nranks * OX_P2P_SLOT;
  L.st = (L.hf + (size_t)2 * nranks * 8 + 255) & ~(size_t)255;
  L.st_stride = (((size_t)n_ghost * OX_MAXC * 8) + 255) & ~(size_t)255;
  L.total = L.st + 2 * L.st_stride;
  if (L.total < 256) L.total = 256;
  return L;
}

extern "C" size_t ox_p2p_window_bytes(int nranks, int64_t n_ghost) { return p2p_layout(nranks, n_ghost).total; }

extern "C" int ox_p2p_window_create(size_t bytes, void **win_dev, char *handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
  if (!win_dev || !handle64 || bytes == 0) OX_FAIL("ox_p2p_window_create: bad argument");
  void *p = nullptr;
  // uncached: stores of a peer become visible without any cache maintenance on this side
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    OX_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  }
  OX_HIP(hipMemset(p, 0, bytes));
  OX_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  OX_HIP(hipIpcGetMemHandle(&h, p));
  memcpy(handle64, &h, sizeof(h));
  *win_dev = p;
  return 0;
}

extern "C" int ox_p2p_window_open(const char *handle64, void **win_dev) {
  if (!handle64 || !win_dev) OX_FAIL("ox_p2p_window_open: bad argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void *p = nullptr;
  OX_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *win_dev = p;
  return 0;
}

extern "C" int ox_p2p_window_close(void *win_dev) {
  if (win_dev) OX_HIP(hipIpcCloseMemHandle(win_dev));
  return 0;
}

extern "C" int ox_p2p_window_free(void *win_dev) {
  if (win_dev) OX_HIP(hipFree(win_dev));
  return 0;
}

template <class T>
static int p2p_upload(T **dst, const T *src, size_t n) {
  OX_HIP(hipMalloc(dst, sizeof(T) * (n ? n : 1)));
  if (n) OX_HIP(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int ox_dist_disable_p2p(ox_dist *d) {
  if (!d || !d->p2p) return 0;
  ox_p2p *q = d->p2p;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < q->n_opened; ++i) (void)hipIpcCloseMemHandle(q->opened[i]);
  free(q->opened);
  (void)hipFree(q->peers_dev);
  (void)hipFree(q->send_off_dev);
  (void)hipFree(q->r_stage);
  (void)hipFree(q->r_off);
  (void)hipFree(q->r_hflag);
  (void)hipFree(q->r_slot);
  (void)hipFree(q->ticket);
  (void)hipFree(q->err_dev);
  if (q->win) (void)hipFree(q->win);
  (void)hipGetLastError();  // tear-down is best effort: leave no stale error for the caller's next HIP call
  free(q);
  d->p2p = nullptr;
  return 0;
}

extern "C" int ox_dist_enable_p2p(ox_dist *d, void *my_win, void *const *rank_wins,
                                  const int64_t *peer_recv_off, const int64_t *peer_n_ghost,
                                  double timeout_s) {
  if (!d || !my_win || !rank_wins) OX_FAIL("ox_dist_enable_p2p: null argument");
  if (d->p2p) OX_FAIL("ox_dist_enable_p2p: already enabled");
  if (d->nranks > 64) OX_FAIL("ox_dist_enable_p2p: at most 64 ranks");
  if (rank_wins[d->rank] != my_win) OX_FAIL("ox_dist_enable_p2p: rank_wins[rank] must be this rank's window");
  ox_p2p *q = static_cast<ox_p2p *>(calloc(1, sizeof(ox_p2p)));
  if (!q) OX_FAIL("ox_dist_enable_p2p: out of memory");
  const int np = d->n_peers, nr = d->nranks;
  q->win = static_cast<char *>(my_win);
  q->n_ghost = d->n_ghost;
  q->win_bytes = p2p_layout(nr, d->n_ghost).total;
  q->opened = static_cast<void **>(calloc(nr, sizeof(void *)));
  for (int r = 0; r < nr; ++r)
    if (r != d->rank && rank_wins[r]) q->opened[q->n_opened++] = rank_wins[r];
  // remote addresses: where THIS rank writes in its peers' windows
  double *r_stage[2 * 64];
  unsigned long long *r_hflag[2 * 64];
  char *r_slot[2 * 64];
  for (int p = 0; p < np; ++p) {
    const int pr = d->peers[p];
    if (pr < 0 || pr >= nr || !rank_wins[pr]) OX_FAIL("ox_dist_enable_p2p: window of peer %d missing", pr);
    const P2pLayout L = p2p_layout(nr, peer_n_ghost[p]);
    char *w = static_cast<char *>(rank_wins[pr]);
    for (int par = 0; par < 2; ++par) {
      r_stage[p * 2 + par] = reinterpret_cast<double *>(w + L.st + par * L.st_stride);
      r_hflag[p * 2 + par] = reinterpret_cast<unsigned long long *>(w + L.hf) + (size_t)par * nr + d->rank;
    }
  }
  for (int r = 0; r < nr; ++r) {
    if (!rank_wins[r]) OX_FAIL("ox_dist_enable_p2p: window of rank %d missing", r);
    char *w = static_cast<char *>(rank_wins[r]);
    for (int par = 0; par < 2; ++par) r_slot[r * 2 + par] = w + ((size_t)par * nr + d->rank) * OX_P2P_SLOT;
  }
  if (p2p_upload(&q->peers_dev, d->peers, np)) return -1;
  if (p2p_upload(&q->send_off_dev, d->send_off, np + 1)) return -1;
  if (p2p_upload(&q->r_stage, r_stage, 2 * np)) return -1;
  if (p2p_upload(&q->r_off, peer_recv_off, np)) return -1;
  if (p2p_upload(&q->r_hflag, r_hflag, 2 * np)) return -1;
  if (p2p_upload(&q->r_slot, r_slot, 2 * nr)) return -1;
  OX_HIP(hipMalloc(&q->ticket, sizeof(unsigned)));
  OX_HIP(hipMemset(q->ticket, 0, sizeof(unsigned)));
  OX_HIP(hipMalloc(&q->err_dev, sizeof(int)));  // device memory: the kernels poll it on every wait
  OX_HIP(hipMemset(q->err_dev, 0, sizeof(int)));
  q->timeout_ticks = (long long)((timeout_s > 0 ? timeout_s : 20.0) * 1e8);  // wall_clock64: 100 MHz
  q->conservative = 1;  // (ox_dist_set_p2p_release(d, 0) selects the fast form)
  OX_HIP(hipDeviceSynchronize());
  d->p2p = q;
  return 0;
}

extern "C" int ox_dist_p2p_timeout(ox_dist *d, double timeout_s) {
  if (!d || !d->p2p) OX_FAIL("ox_dist_p2p_timeout: the plan has no xGMI transport");
  d->p2p->timeout_ticks = (long long)((timeout_s > 0 ? timeout_s : 20.0) * 1e8);
  return 0;
}

extern "C" int ox_dist_set_overlap(ox_dist *d, int overlap) {
  if (!d) OX_FAIL("ox_dist_set_overlap: null plan");
  d->overlap = overlap < 0 ? -1 : (overlap ? 1 : 0);
  return 0;
}

extern "C" int ox_dist_set_p2p_release(ox_dist *d, int conservative) {
  if (!d) OX_FAIL("ox_dist_set_p2p_release: null plan");
  if (!d->p2p) OX_FAIL("ox_dist_set_p2p_release: the plan has no xGMI-window transport (ox_dist_enable_p2p first)");
  d->p2p->conservative = conservative ? 1 : 0;
  return 0;
}

extern "C" int ox_dist_status(const ox_dist *d) {
  if (!d || !d->p2p) return 0;
  int e = 0;  // callers have drained the stream: a blocking 4-byte copy
  OX_HIP(hipMemcpy(&e, d->p2p->err_dev, sizeof(int), hipMemcpyDeviceToHost));
  if (e) OX_FAIL("xGMI transport: a wait for a peer rank timed out (rank %d)", d->rank);
  return 0;
}

// 16-byte system-scope accesses to a window (no builtin for them): write-through store, cache-bypassing load
typedef unsigned ox_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ox_store16_sys(double *dst, double a, double b) {
  ox_u4 v;
  v.x = (unsigned)__double_as_longlong(a);
  v.y = (unsigned)((unsigned long long)__double_as_longlong(a) >> 32);
  v.z = (unsigned)__double_as_longlong(b);
  v.w = (unsigned)((unsigned long long)__double_as_longlong(b) >> 32);
  // (s_nop 1: the compiler does not see a VMEM store inside inline asm, so it cannot place the wait states a store of
  // more than 64 bits needs before its data registers are written again)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void ox_load16_sys(const double *src, double &a, double &b) {
  ox_u4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
  a = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
  b = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
}

// gather the interface values and store them into the neighbours' windows; the last block to
// finish raises this rank's flag in every neighbour's window.  A thread moves TWO consecutive values of the flattened
// [entry][component] list: one 16-byte store where both go to the same neighbour at a 16-byte aligned place (every
// system-scope store into the uncached window is a fabric write of its own, whatever its width).
__global__ __launch_bounds__(256) void k_halo_push(const double *__restrict__ x,
                                                   const int32_t *__restrict__ idx, int64_t ns, int nc,
                                                   const int64_t *__restrict__ send_off, int n_peers,
                                                   double *const *__restrict__ r_stage,
                                                   const int64_t *__restrict__ r_off,
                                                   unsigned long long *const *__restrict__ r_hflag,
                                                   int parity, unsigned long long seq, unsigned *ticket, int conservative) {
  const int64_t e0 = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x), tot = ns * nc;
  if (e0 < tot) {
    double v[2];
    double *dst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t e = e0 + j < tot ? e0 + j : e0;  // (an odd total: the last thread's second value is its first again)
      const int64_t k = e / nc;
      const int c = (int)(e - k * nc);
      int p = 0;
      while (p + 1 < n_peers && k >= send_off[p + 1]) ++p;
      v[j] = x[(int64_t)idx[k] * nc + c];
      dst[j] = r_stage[p * 2 + parity] + (r_off[p] + (k - send_off[p])) * nc + c;
    }
    if (e0 + 1 < tot && dst[1] == dst[0] + 1 && (reinterpret_cast<size_t>(dst[0]) & 15) == 0) {
      ox_store16_sys(dst[0], v[0], v[1]);
    } else {
      __hip_atomic_store(dst[0], v[0], __ATOMIC_RELAXED, OX_SYS);
      if (e0 + 1 < tot) __hip_atomic_store(dst[1], v[1], __ATOMIC_RELAXED, OX_SYS);
    }
  }
  // The payload stores are system-scope write-through stores into an uncached window: nothing of them sits in a cache.
  // Every storing wave waits for its stores to be acknowledged, the block barrier orders them in front of thread 0's
  // agent-scope acq_rel ticket, and the block that arrives last fences ONCE at system scope before it raises the flags.
  // (Rounds 1-4 had every thread of every block run __threadfence_system() here: a write-back and an invalidate of the
  // XCD's L2 per block -- 100 us for the 2.4 MB velocity halo of a 128^3 x 8 rank, and the mat-vec around it lost its
  // cached operands; round 5, found by timing the self-loop plans: tools/predict_scaling.py --transport p2p.)
  // That form has only ever run with every rank on ONE device.  CONSERVATIVE plans (ox_dist_set_p2p_release, the
  // default until a run between two GPUs has passed the halo self-test and a bit-exact all-reduce) keep the per-thread
  // system fence of rounds 1-4 behind the payload stores and raise the flags with system-scope RELEASE stores.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (conservative) __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();  // the release of the launch
      if (conservative) {
        for (int p = 0; p < n_peers; ++p) __hip_atomic_store(r_hflag[p * 2 + parity], seq, __ATOMIC_RELEASE, OX_SYS);
      } else {  // fast form: the flags behind the one fence are relaxed stores
        for (int p = 0; p < n_peers; ++p) __hip_atomic_store(r_hflag[p * 2 + parity], seq, __ATOMIC_RELAXED, OX_SYS);
      }
    }
  }
}

// wait for every neighbour's flag, then move the staged values into the ghost block
__global__ __launch_bounds__(256) void k_halo_pull(double *__restrict__ ghost, int64_t n, const double *stage,
                                                   const unsigned long long *hflag,
                                                   const int32_t *__restrict__ peers, int n_peers,
                                                   unsigned long long seq, long long timeout_ticks, int *err) {
  __shared__ int failed;
  if (threadIdx.x == 0) failed = 0;
  __syncthreads();
  if ((int)threadIdx.x < n_peers && !ox_p2p_wait(hflag + peers[threadIdx.x], seq, timeout_ticks, err)) failed = 1;
  __syncthreads();
  if (failed) return;  // a peer never delivered: leave the ghost block alone (the sticky flag fails the host call)
  const int64_t stride = (int64_t)gridDim.x * 256, n2 = n >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {  // (the staging area is 256-byte aligned)
    double a, b;
    ox_load16_sys(stage + 2 * i, a, b);
    ghost[2 * i] = a;
    ghost[2 * i + 1] = b;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) ghost[n - 1] = __hip_atomic_load(stage + n - 1, __ATOMIC_RELAXED, OX_SYS);
}

// all-reduce (sum) of n <= 15 doubles: thread r talks to rank r
__global__ __launch_bounds__(1024) void k_allreduce_p2p(double *buf, int n, ox_p2p_ar a) {
  __shared__ double stage[64][OX_P2P_MAXV + 1];
  __shared__ double vals[OX_P2P_MAXV + 1];
  if ((int)threadIdx.x < n) vals[threadIdx.x] = buf[threadIdx.x];
  __syncthreads();
  ox_p2p_allreduce_block(vals, n, a, stage);
  if ((int)threadIdx.x < n) buf[threadIdx.x] = vals[threadIdx.x];
}

ox_p2p_ar ox_p2p_next_allreduce(const ox_dist *d) {
  ox_p2p *q = d->p2p;
  ox_p2p_ar a;
  a.seq = ++q->aseq;
  a.parity = (int)(a.seq & 1);
  a.r_slot = q->r_slot;
  a.my_slots = q->win;
  a.nranks = d->nranks;
  a.timeout_ticks = q->timeout_ticks;
  a.err = q->err_dev;
  a.conservative = q->conservative;
  return a;
}

static int p2p_halo_push(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  ox_p2p *q = d->p2p;
  const unsigned long long seq = ++q->hseq;
  const int parity = (int)(seq & 1);
  const int64_t ns = d->send_off[d->n_peers];
  const int64_t tot = (ns * ncomp + 1) / 2;  // two values per thread
  const unsigned nblk = (unsigned)(tot > 0 ? (tot + 255) / 256 : 1);
  hipLaunchKernelGGL(k_halo_push, dim3(nblk), dim3(256), 0, st, x, d->send_idx, ns, ncomp, q->send_off_dev,
                     d->n_peers, q->r_stage, q->r_off, q->r_hflag, parity, seq, q->ticket, q->conservative);
  OX_LAUNCH_CHECK();
  return 0;
}

static int p2p_halo_pull(const ox_dist *d, double *x, int ncomp, hipStream_t st) {  // of the latest push
  ox_p2p *q = d->p2p;
  const unsigned long long seq = q->hseq;
  const int parity = (int)(seq & 1);
  const P2pLayout L = p2p_layout(d->nranks, d->n_ghost);
  const int64_t ng = d->n_ghost * ncomp;
  unsigned nb2 = (unsigned)((ng + 255) / 256);
  if (nb2 < 1) nb2 = 1;
  if (nb2 > 1024) nb2 = 1024;
  hipLaunchKernelGGL(k_halo_pull, dim3(nb2), dim3(256), 0, st, x + d->n_owned * ncomp, ng,
                     reinterpret_cast<const double *>(q->win + L.st + parity * L.st_stride),
                     reinterpret_cast<const unsigned long long *>(q->win + L.hf) + (size_t)parity * d->nranks,
                     q->peers_dev, d->n_peers, seq, q->timeout_ticks, q->err_dev);
  OX_LAUNCH_CHECK();
  return 0;
}

static int p2p_halo_forward(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  if (p2p_halo_push(d, x, ncomp, st)) return -1;
  return p2p_halo_pull(d, x, ncomp, st);
}

static int p2p_allreduce(const ox_dist *d, double *buf, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_allreduce_p2p, dim3(1), dim3(ox_p2p_ar_threads(d->nranks)), 0, st, buf, n, ox_p2p_next_allreduce(d));
  OX_LAUNCH_CHECK();
  return 0;
}

extern "C" int ox_dist_destroy(ox_dist *d) {
  if (!d) return 0;
  ox_dist_disable_p2p(d);
  if (d->send_buf) (void)hipFree(d->send_buf);
  if (d->side) {
    (void)hipStreamSynchronize(d->side);
    (void)hipEventDestroy(d->ev_begin);
    (void)hipEventDestroy(d->ev_done);
    (void)hipStreamDestroy(d->side);
  }
  free(d->peers);
  free(d->send_off);
  free(d->recv_off);
  free(d);
  return 0;
}

__global__ void k_pack(const double *__restrict__ x, const int32_t *__restrict__ idx, int64_t n,
                       int nc, double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * nc) return;
  const int64_t k = i / nc;
  const int c = (int)(i - k * nc);
  buf[i] = x[(int64_t)idx[k] * nc + c];
}

static int halo_forward_rccl_or_cb(const ox_dist *d, double *x, int ncomp, hipStream_t st);

int ox_halo_forward_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  if (!d || d->n_peers == 0) return 0;
  if (ox_prof_on) ox_prof_start(OX_TAG_HALO, st, ncomp);
  const int rc = d->p2p ? p2p_halo_forward(d, x, ncomp, st) : halo_forward_rccl_or_cb(d, x, ncomp, st);
  if (ox_prof_on) ox_prof_stop(st);
  return rc;
}

int ox_halo_begin_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  if (!d || d->n_peers == 0) return 0;
  if (d->p2p) return p2p_halo_push(d, x, ncomp, st);
  if (d->halo_cb || !d->comm) return halo_forward_rccl_or_cb(d, x, ncomp, st);  // synchronous transport: all of it now
  // RCCL: pack + grouped send/recv on the side stream, behind everything enqueued on `st` so far
  ox_dist *m = const_cast<ox_dist *>(d);
  if (!m->side) {
    OX_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
    OX_HIP(hipEventCreateWithFlags(&m->ev_begin, hipEventDisableTiming));
    OX_HIP(hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming));
  }
  OX_HIP(hipEventRecord(m->ev_begin, st));
  OX_HIP(hipStreamWaitEvent(m->side, m->ev_begin, 0));
  if (halo_forward_rccl_or_cb(d, x, ncomp, m->side)) return -1;
  OX_HIP(hipEventRecord(m->ev_done, m->side));
  return 0;
}

int ox_halo_end_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  if (!d || d->n_peers == 0) return 0;
  if (d->p2p) return p2p_halo_pull(d, x, ncomp, st);
  if (d->halo_cb || !d->comm) return 0;
  OX_HIP(hipStreamWaitEvent(st, d->ev_done, 0));
  return 0;
}

static int halo_forward_rccl_or_cb(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  if (!d->comm && !d->halo_cb) OX_FAIL("ox_halo_forward: the plan has no transport (RCCL communicator, xGMI windows or callbacks)");
  ncclComm_t comm = static_cast<ncclComm_t>(d->comm);
  const int64_t ns = d->send_off[d->n_peers];
  if (ns > 0) {
    const int64_t tot = ns * ncomp;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, d->send_idx,
                       ns, ncomp, d->send_buf);
    OX_LAUNCH_CHECK();
  }
  if (d->halo_cb) {
    OX_HIP(hipStreamSynchronize(st));
    if (d->halo_cb(d->user, d->send_buf, x + d->n_owned * ncomp, ncomp)) OX_FAIL("halo callback failed");
    return 0;
  }
  OX_NCCL(ncclGroupStart());
  for (int p = 0; p < d->n_peers; ++p) {
    const int64_t sc = d->send_off[p + 1] - d->send_off[p];
    const int64_t rc = d->recv_off[p + 1] - d->recv_off[p];
    if (sc > 0)
      OX_NCCL(ncclSend(d->send_buf + d->send_off[p] * ncomp, (size_t)(sc * ncomp), ncclDouble,
                       d->peers[p], comm, st));
    if (rc > 0)
      OX_NCCL(ncclRecv(x + (d->n_owned + d->recv_off[p]) * ncomp, (size_t)(rc * ncomp), ncclDouble,
                       d->peers[p], comm, st));
  }
  OX_NCCL(ncclGroupEnd());
  return 0;
}

int ox_allreduce_impl(const ox_dist *d, double *buf, int n, hipStream_t st) {
  if (!d || d->nranks == 1) return 0;
  if (d->p2p && n <= OX_P2P_MAXV) return p2p_allreduce(d, buf, n, st);
  if (!d->comm && !d->allreduce_cb) OX_FAIL("ox_allreduce_sum: the plan has no transport for %d values", n);
  if (d->allreduce_cb) {
    OX_HIP(hipStreamSynchronize(st));
    if (d->allreduce_cb(d->user, buf, n)) OX_FAIL("allreduce callback failed");
    return 0;
  }
  OX_NCCL(ncclAllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, static_cast<ncclComm_t>(d->comm), st));
  return 0;
}

extern "C" int ox_halo_forward(const ox_dist *d, double *x, int ncomp, void *stream) {
  if (!x) OX_FAIL("ox_halo_forward: null vector");
  if (ncomp < 1 || ncomp > OX_MAXC) OX_FAIL("ox_halo_forward: ncomp=%d", ncomp);
  return ox_halo_forward_impl(d, x, ncomp, ox_stream(stream));
}

extern "C" int ox_allreduce_sum(const ox_dist *d, double *buf_dev, int n, void *stream) {
  return ox_allreduce_impl(d, buf_dev, n, ox_stream(stream));
}
