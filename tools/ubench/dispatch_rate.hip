// How fast can the chip START waves?  Empty / sleeping / one-load kernels on the grid shapes of the
// pressure SpMV (8392 blocks x 256 threads at 128^3): the floor any one-slice-per-wave kernel pays.
// hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dispatch_rate && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDS>
__global__ void k_empty(double *out) {
  __shared__ double buf[LDS > 0 ? LDS : 1];
  if (LDS > 0) buf[threadIdx.x % LDS] = 1.0;
  if (blockIdx.x == 0x7fffffff) out[threadIdx.x] = buf[0];
}
// each wave sleeps ~cycles (s_sleep 1 = 64 cycles)
__global__ void k_sleep(double *out, int n64) {
  for (int i = 0; i < n64; ++i) __builtin_amdgcn_s_sleep(1);
  if (blockIdx.x == 0x7fffffff) out[threadIdx.x] = 1.0;
}
// each wave: CH dependent coalesced 512-B loads from a large array, then one 512-B store
template <int CH>
__global__ void k_chain(const long long *__restrict__ next, double *__restrict__ out, long long n) {
  const int lane = threadIdx.x & 63;
  long long w = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  long long idx = (w * 64) % n;
  long long v = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    v = next[idx + lane];         // value = start of the next 64-element block
    idx = __builtin_amdgcn_readfirstlane((int)(v & 0x7fffffff));
  }
  out[w * 64 + lane] = (double)v;
}
// K independent coalesced 512-B loads per wave (no chain), one store
template <int K>
__global__ void k_par(const double *__restrict__ a, double *__restrict__ out, long long n) {
  const int lane = threadIdx.x & 63;
  long long w = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  double s = 0;
#pragma unroll
  for (int c = 0; c < K; ++c) s += a[((w * K + c) * 64) % n + lane];
  out[w * 64 + lane] = s;
}

// K independent loads of T per lane (64 x sizeof(T) contiguous bytes per wave instruction)
template <class T, int K>
__global__ void k_part(const T *__restrict__ a, double *__restrict__ out, long long n) {
  const int lane = threadIdx.x & 63;
  long long w = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const T v = a[((w * K + c) * 64) % n + lane];
    const unsigned *p = reinterpret_cast<const unsigned *>(&v);
    s += p[0];
    if (sizeof(T) >= 8) s += p[sizeof(T) >= 8 ? 1 : 0];
    if (sizeof(T) >= 16) s += p[sizeof(T) >= 16 ? 2 : 0] + p[sizeof(T) >= 16 ? 3 : 0];
  }
  out[w * 64 + lane] = (double)s;
}
// the same with a lane-permuted (gather-like) address: lane l reads element (l * 37) % 64 of the block
template <class T, int K>
__global__ void k_gath(const T *__restrict__ a, double *__restrict__ out, long long n) {
  const int lane = ((threadIdx.x & 63) * 37) & 63;
  long long w = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const T v = a[((w * K + c) * 64) % n + lane];
    const unsigned *p = reinterpret_cast<const unsigned *>(&v);
    s += p[0];
    if (sizeof(T) >= 8) s += p[sizeof(T) >= 8 ? 1 : 0];
    if (sizeof(T) >= 16) s += p[sizeof(T) >= 16 ? 2 : 0] + p[sizeof(T) >= 16 ? 3 : 0];
  }
  out[w * 64 + lane] = (double)s;
}

// K loads of 16 B per lane at byte address block + off + lane * stride (pair-gather shapes)
template <int K>
__global__ void k_ovl(const char *__restrict__ a, double *__restrict__ out, long long nbytes, int stride, int off) {
  typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(8)));
  const int lane = threadIdx.x & 63;
  long long w = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const u4 v = *reinterpret_cast<const u4 *>(a + ((w * K + c) * 1024) % nbytes + off + lane * stride);
    s += v.x + v.y + v.z + v.w;
  }
  out[w * 64 + lane] = (double)s;
}
template <int K>
__global__ void k_ovl8(const char *__restrict__ a, double *__restrict__ out, long long nbytes, int stride, int off) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  long long w = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const u2 v = *reinterpret_cast<const u2 *>(a + ((w * K + c) * 1024) % nbytes + off + lane * stride);
    s += v.x + v.y;
  }
  out[w * 64 + lane] = (double)s;
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const long long waves = 33568;
  double *out; CK(hipMalloc(&out, (waves + 64) * 64 * 8 * 8));
  const long long n = 1ll << 24;  // 128 MB of int64 / double
  long long *nx; CK(hipMalloc(&nx, n * 8));
  double *a; CK(hipMalloc(&a, n * 8));
  CK(hipMemset(a, 0, n * 8));
  {
    std::vector<long long> h(n);
    unsigned long long s = 88172645463325252ull;
    for (long long b = 0; b < n / 64; ++b) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const long long t = (long long)(s % (unsigned long long)(n / 64)) * 64;
      for (int l = 0; l < 64; ++l) h[b * 64 + l] = t;
    }
    CK(hipMemcpy(nx, h.data(), n * 8, hipMemcpyHostToDevice));
  }
  const int R = 200;
  printf("waves per launch: %lld\n", waves);
  printf("empty, 64-thread blocks   : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty<0>, dim3(waves), dim3(64), 0, 0, out); }, R));
  printf("empty, 256-thread blocks  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty<0>, dim3(waves / 4), dim3(256), 0, 0, out); }, R));
  printf("empty, 1024-thread blocks : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty<0>, dim3(waves / 16), dim3(1024), 0, 0, out); }, R));
  printf("empty, 256 thr, 2 KB LDS  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty<256>, dim3(waves / 4), dim3(256), 0, 0, out); }, R));
  printf("empty, 256 thr, 8 KB LDS  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty<1024>, dim3(waves / 4), dim3(256), 0, 0, out); }, R));
  printf("empty, 256 thr, 1 block   : %7.2f us (launch floor)\n", timeit([&] { hipLaunchKernelGGL(k_empty<0>, dim3(1), dim3(256), 0, 0, out); }, R));
  for (int n64 : {8, 16, 32, 64, 128, 256})
    printf("sleep %5d cycles, 256 thr: %7.2f us\n", n64 * 64, timeit([&] { hipLaunchKernelGGL(k_sleep, dim3(waves / 4), dim3(256), 0, 0, out, n64); }, R));
  printf("chain of 1 load + store   : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<1>, dim3(waves / 4), dim3(256), 0, 0, nx, out, n); }, R));
  printf("chain of 2 loads + store  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<2>, dim3(waves / 4), dim3(256), 0, 0, nx, out, n); }, R));
  printf("chain of 3 loads + store  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<3>, dim3(waves / 4), dim3(256), 0, 0, nx, out, n); }, R));
  printf("chain of 4 loads + store  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<4>, dim3(waves / 4), dim3(256), 0, 0, nx, out, n); }, R));
  printf("chain of 6 loads + store  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<6>, dim3(waves / 4), dim3(256), 0, 0, nx, out, n); }, R));
  for (long long m : {1ll << 14, 1ll << 17, 1ll << 19, 1ll << 21, 1ll << 23}) {  // array of m doubles
    printf("16 parallel loads, %6.1f MB array: %7.2f us | 32 loads: %7.2f us\n", m * 8 / 1e6,
           timeit([&] { hipLaunchKernelGGL(k_par<16>, dim3(waves / 4), dim3(256), 0, 0, a, out, m); }, R),
           timeit([&] { hipLaunchKernelGGL(k_par<32>, dim3(waves / 4), dim3(256), 0, 0, a, out, m); }, R));
  }
  printf("16 loads of ONE 512-B block (L1 hits): %7.2f us | 32 loads: %7.2f us\n",
         timeit([&] { hipLaunchKernelGGL(k_par<16>, dim3(waves / 4), dim3(256), 0, 0, a, out, 64ll); }, R),
         timeit([&] { hipLaunchKernelGGL(k_par<32>, dim3(waves / 4), dim3(256), 0, 0, a, out, 64ll); }, R));
  {
    const long long m = 1ll << 17;  // 1 MB of doubles: L2-resident
#define ROW(T, name)                                                                                          \
    printf("16 loads of %-8s per lane (L2-resident): %7.2f us | permuted lanes: %7.2f us | one block (L1): %7.2f us\n", name, \
           timeit([&] { hipLaunchKernelGGL((k_part<T, 16>), dim3(waves / 4), dim3(256), 0, 0, (const T *)a, out, m * 8 / (long long)sizeof(T)); }, R), \
           timeit([&] { hipLaunchKernelGGL((k_gath<T, 16>), dim3(waves / 4), dim3(256), 0, 0, (const T *)a, out, m * 8 / (long long)sizeof(T)); }, R), \
           timeit([&] { hipLaunchKernelGGL((k_part<T, 16>), dim3(waves / 4), dim3(256), 0, 0, (const T *)a, out, 64ll); }, R));
    ROW(unsigned short, "ushort")
    ROW(unsigned, "dword")
    ROW(uint2, "dwordx2")
    ROW(uint4, "dwordx4")
  }
  {
    const long long nb = 1ll << 20;  // L2-resident
    const char *ac = (const char *)a;
    printf("16 x dwordx2, lane stride 8             : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl8<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 8, 0); }, R));
    printf("16 x dwordx2, lane stride 8, +8 B (mod 16): %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl8<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 8, 8); }, R));
    printf("16 x dwordx2, lane stride 8, +40 B       : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl8<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 8, 40); }, R));
    printf("16 x dwordx2, lane stride 24             : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl8<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 24, 0); }, R));
    printf("16 x dwordx4, lane stride 16 (aligned)   : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 16, 0); }, R));
    printf("16 x dwordx4, lane stride 16, +8 B       : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 16, 8); }, R));
    printf("16 x dwordx4, lane stride 8 (overlapping): %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 8, 0); }, R));
    printf("16 x dwordx4, lane stride 8, +8 B        : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 8, 8); }, R));
    printf("16 x dwordx4, lane stride 8, +40 B       : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 8, 40); }, R));
    printf("16 x dwordx4, lane stride 24             : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 24, 0); }, R));
    printf("16 x dwordx4, lane stride 24, +8 B       : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_ovl<16>, dim3(waves / 4), dim3(256), 0, 0, ac, out, nb, 24, 8); }, R));
  }
  printf("4 parallel loads + store  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_par<4>, dim3(waves / 4), dim3(256), 0, 0, a, out, n); }, R));
  printf("8 parallel loads + store  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_par<8>, dim3(waves / 4), dim3(256), 0, 0, a, out, n); }, R));
  printf("16 parallel loads + store : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_par<16>, dim3(waves / 4), dim3(256), 0, 0, a, out, n); }, R));
  printf("32 parallel loads + store : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_par<32>, dim3(waves / 4), dim3(256), 0, 0, a, out, n); }, R));
  CK(hipDeviceSynchronize());
  return 0;
}
