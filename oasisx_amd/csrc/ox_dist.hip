// Mesh-partitioned runs: one process per GPU, RCCL over xGMI.  The halo exchange
// (DOLFINx scatter_forward, reference fracstep.py:453,497,502,551,632,655; ksp.py:77) is a
// pack kernel + one grouped ncclSend/ncclRecv per neighbour straight into the ghost block of
// the vector; Krylov scalars are merged into one small ncclAllReduce per synchronisation point.
#include <rccl/rccl.h>

#include <stdlib.h>

#include "ox_kernels.h"

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

extern "C" int ox_comm_destroy(void *comm) {
  if (comm) ncclCommDestroy(static_cast<ncclComm_t>(comm));
  return 0;
}

extern "C" int ox_dist_create(void *comm, int rank, int nranks, int n_peers, const int32_t *peers,
                              const int64_t *send_off, const int32_t *send_idx_dev,
                              const int64_t *recv_off, int64_t n_owned, int64_t n_ghost,
                              ox_dist **out) {
  if (!comm || !out || nranks < 1 || rank < 0 || rank >= nranks) OX_FAIL("ox_dist_create: bad argument");
  ox_dist *d = static_cast<ox_dist *>(calloc(1, sizeof(ox_dist)));
  if (!d) OX_FAIL("ox_dist_create: out of memory");
  d->comm = comm;
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

extern "C" int ox_dist_destroy(ox_dist *d) {
  if (!d) return 0;
  if (d->send_buf) (void)hipFree(d->send_buf);
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

int ox_halo_forward_impl(const ox_dist *d, double *x, int ncomp, hipStream_t st) {
  if (!d || d->n_peers == 0) return 0;
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
