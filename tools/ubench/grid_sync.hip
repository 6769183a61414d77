// What do the synchronisation points of a PERSISTENT Krylov iteration cost on gfx950?  One 1024-thread block per CU
// (256 blocks), all resident; per "iteration" the pattern of a conjugate-gradient step whose vectors live in registers:
//   R: an all-reduce of NV doubles -- every block publishes its partial sums as tagged 8-byte granules ({epoch, half of
//      the double}: the data is the flag), one wave per block sweeps all 256 x 2 NV granules until every tag matches,
//      sums them in a fixed order (the same bits in every block);
//   P: publish a slab of the search direction (write-through stores, every wave drained, block barrier), the same
//      granule sweep as a barrier, ONE agent-scope acquire per block, then plain loads of OTHER blocks' slabs (checked).
// Modes: 0 = R only, 1 = R R P (standard CG), 2 = R P (merged-reduction CG), 3 = P only.
// Every spin is bounded (time-out -> error word -> every block leaves).
//   hipcc --offload-arch=gfx950 -O3 grid_sync.hip -o grid_sync.bin && ./grid_sync.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
#define AGENT __HIP_MEMORY_SCOPE_AGENT

constexpr int NB = 256;       // blocks = CUs
constexpr int NT = 1024;      // threads per block
constexpr int NV = 2;         // doubles per all-reduce
constexpr int ROWS = 8;       // rows of the slab per thread (8 x 1024 x 8 B = 64 KB per block)

struct Sync {
  u64 *gran;        // [2][NB][2 * NV] tagged granules of the all-reduces
  u64 *bar;         // [2][NB] tagged granules of the barriers
  int *err;
  long long timeout_ticks;
};

// one wave: wait until the N granules of every block carry `epoch`; v[k] = the 32-bit payloads of this lane's blocks
template <int N>
__device__ __forceinline__ bool sweep(const u64 *g, unsigned epoch, unsigned (*v)[N], const Sync &S) {
  const int lane = threadIdx.x & 63;
  const long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NB / 64; ++j) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const u64 x = __hip_atomic_load(g + (size_t)(lane + 64 * j) * N + k, __ATOMIC_RELAXED, AGENT);
        v[j][k] = (unsigned)x;
        ok &= (unsigned)(x >> 32) == epoch;
      }
    }
    if (__all(ok)) return true;
    if (__hip_atomic_load(S.err, __ATOMIC_RELAXED, AGENT)) return false;
    if (wall_clock64() - t0 > S.timeout_ticks) {
      __hip_atomic_store(S.err, 1, __ATOMIC_RELAXED, AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// all-reduce of NV doubles: s[] in (this block's partial sums, valid in thread 0), out in LDS sh[0..NV)
__device__ __forceinline__ bool allreduce(double *sh, int *sh_ok, unsigned epoch, const Sync &S) {
  u64 *g = S.gran + (size_t)(epoch & 1) * NB * 2 * NV;
  if (threadIdx.x < 2 * NV) {
    const u64 bits = (u64)__double_as_longlong(sh[threadIdx.x >> 1]);
    const unsigned half = (threadIdx.x & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
    __hip_atomic_store(g + (size_t)blockIdx.x * 2 * NV + threadIdx.x, ((u64)epoch << 32) | half, __ATOMIC_RELAXED, AGENT);
  }
  if (threadIdx.x < 64) {
    unsigned v[NB / 64][2 * NV];
    const bool ok = sweep<2 * NV>(g, epoch, v, S);
    if (threadIdx.x == 0) *sh_ok = ok;
    double t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      t[i] = 0.0;
#pragma unroll
      for (int j = 0; j < NB / 64; ++j)
        t[i] += __longlong_as_double((long long)(((u64)v[j][2 * i + 1] << 32) | v[j][2 * i]));
      // fixed butterfly: the same bits in every block
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t[i] += __shfl_xor(t[i], o);
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) sh[i] = t[i];
    }
  }
  __syncthreads();
  return *sh_ok != 0;
}

// barrier behind write-through stores: every wave has drained (caller), then one granule per block, sweep, acquire
__device__ __forceinline__ bool barrier_acquire(unsigned epoch, const Sync &S, int *sh_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  u64 *g = S.bar + (size_t)(epoch & 1) * NB;
  if (threadIdx.x == 0) __hip_atomic_store(g + blockIdx.x, ((u64)epoch << 32) | 1u, __ATOMIC_RELAXED, AGENT);
  if (threadIdx.x < 64) {
    unsigned v[NB / 64][1];
    const bool ok = sweep<1>(g, epoch, v, S);
    if (threadIdx.x == 0) {
      *sh_ok = ok;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return *sh_ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(NT) void k_sync(Sync S, double *__restrict__ p, int iters, double *out, int *bad) {
  __shared__ double sh[NV];
  __shared__ int sh_ok;
  unsigned epoch = 0;
  double acc = 0.0;
  int nbad = 0;
  const size_t slab = (size_t)NT * ROWS;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 1 || MODE == 2) {
      if (threadIdx.x == 0) { sh[0] = (double)(blockIdx.x + 1) * (it + 1); sh[1] = 1.0; }
      __syncthreads();
      if (!allreduce(sh, &sh_ok, ++epoch, S)) break;
      acc += sh[0] + sh[1];
      __syncthreads();
    }
    if (MODE == 1) {
      if (threadIdx.x == 0) { sh[0] = acc * 1e-9; sh[1] = 2.0; }
      __syncthreads();
      if (!allreduce(sh, &sh_ok, ++epoch, S)) break;
      acc += sh[0];
      __syncthreads();
    }
    if (MODE == 1 || MODE == 2 || MODE == 3) {
      // the slab of this block, write-through
#pragma unroll
      for (int k = 0; k < ROWS; ++k)
        __hip_atomic_store(p + blockIdx.x * slab + (size_t)k * NT + threadIdx.x, (double)(it + 1) + 1e-3 * blockIdx.x,
                           __ATOMIC_RELAXED, AGENT);
      if (!barrier_acquire(++epoch, S, &sh_ok)) break;
      // plain loads of two other blocks' slabs (one same-XCD + 8, one neighbour)
      const int b1 = (blockIdx.x + 1) % NB, b2 = (blockIdx.x + 8) % NB;
#pragma unroll
      for (int k = 0; k < ROWS; k += 4) {
        const double v1 = p[b1 * slab + (size_t)k * NT + threadIdx.x], v2 = p[b2 * slab + (size_t)k * NT + threadIdx.x];
        nbad += v1 != (double)(it + 1) + 1e-3 * b1;
        nbad += v2 != (double)(it + 1) + 1e-3 * b2;
        acc += v1 + v2;
      }
      // nobody may overwrite a slab before everybody has read it: the NEXT iteration's first synchronisation point
      // orders that in modes 1 and 2; mode 3 needs a barrier of its own
      if (MODE == 3) {
        if (!barrier_acquire(++epoch, S, &sh_ok)) break;
      }
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

template <int MODE>
static int run(const char *what, Sync S, double *p, double *out, int *bad, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(S.gran, 0, sizeof(u64) * 2 * NB * 2 * NV));
    CK(hipMemset(S.bar, 0, sizeof(u64) * 2 * NB));
    CK(hipMemset(S.err, 0, sizeof(int)));
    CK(hipMemset(bad, 0, sizeof(int)));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_sync<MODE>, dim3(NB), dim3(NT), 0, 0, S, p, iters, out, bad);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    int err = 0, nb = 0;
    CK(hipMemcpy(&err, S.err, sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&nb, bad, sizeof(int), hipMemcpyDeviceToHost));
    if (err || nb) {
      printf("%s: TIME-OUT %d, stale reads %d\n", what, err, nb);
      return 1;
    }
  }
  printf("%-58s %7.2f us per iteration (%d iterations, best of 4)\n", what, best * 1e3 / iters, iters);
  return 0;
}

int main() {
  Sync S;
  CK(hipMalloc(&S.gran, sizeof(u64) * 2 * NB * 2 * NV));
  CK(hipMalloc(&S.bar, sizeof(u64) * 2 * NB));
  CK(hipMalloc(&S.err, sizeof(int)));
  S.timeout_ticks = 100000000LL / 50;  // 20 ms of the 100 MHz clock
  double *p, *out;
  int *bad;
  CK(hipMalloc(&p, sizeof(double) * (size_t)NB * NT * ROWS));
  CK(hipMalloc(&out, sizeof(double) * NB));
  CK(hipMalloc(&bad, sizeof(int)));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sync<1>, NT, 0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs, %d block(s) of %d threads per CU by the occupancy query\n", prop.name, prop.multiProcessorCount, occ, NT);
  if (prop.multiProcessorCount < NB || occ < 1) { printf("grid would not be resident\n"); return 1; }
  const int iters = 2000;
  int rc = 0;
  rc |= run<0>("R   (all-reduce of 2 doubles)", S, p, out, bad, iters);
  rc |= run<3>("P P (publish 64 KB/block + barrier + acquire, twice)", S, p, out, bad, iters);
  rc |= run<2>("R P (merged-reduction CG)", S, p, out, bad, iters);
  rc |= run<1>("R R P (standard CG)", S, p, out, bad, iters);
  return rc;
}
