// Streaming-read microbenchmark: one wave reads a contiguous "slice" of W steps; per step a lane
// loads VB bytes of "values" and CB bytes of "columns" (separate arrays), coalesced over the wave.
// Question: does the bytes-per-lane-per-load shape change the achievable HBM read rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1);} } while (0)

template <int VD /* doubles per lane per step */, int CI /* ints per lane per step */, int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ vals, const int* __restrict__ cols,
                                                int steps, int nslices, double* out) {
  const int ngroups = (nslices + 3) >> 2;
  const int chunk = (ngroups + 7) >> 3;
  const int g = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunk || g >= ngroups) return;
  const int slice = g * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= nslices) return;
  const double* vp = vals + ((size_t)slice * steps * 64 + lane) * VD;
  const int* cp = cols + ((size_t)slice * steps * 64 + lane) * CI;
  double acc = 0.0;
  int iacc = 0;
#pragma unroll UNROLL
  for (int k = 0; k < steps; ++k) {
#pragma unroll
    for (int j = 0; j < VD; j += 2) {
      const double2 v = *reinterpret_cast<const double2*>(vp + (size_t)k * 64 * VD + j);
      acc += v.x + v.y;
    }
#pragma unroll
    for (int j = 0; j < CI; j += 2) {
      const int2 c = *reinterpret_cast<const int2*>(cp + (size_t)k * 64 * CI + j);
      iacc += c.x ^ c.y;
    }
  }
  if (acc == 123.456 && iacc == 7) out[0] = acc;  // keep the loads alive
}

template <int VD, int CI, int UNROLL>
void run(const char* name, double* vals, int* cols, double* out, size_t total_slots) {
  // same total bytes for every shape: slots of (8 B value + 4 B column)
  const int steps_slots = 64;                      // entries per row ("width")
  const int steps = steps_slots / VD;              // steps per slice
  const int nslices = (int)(total_slots / (64 * (size_t)steps_slots));
  const int ngroups = (nslices + 3) / 4;
  const int nblk = (ngroups + 7) & ~7;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<VD, CI, UNROLL>), dim3(nblk), dim3(256), 0, 0, vals, cols, steps, nslices, out);
  CK(hipEventRecord(a));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<VD, CI, UNROLL>), dim3(nblk), dim3(256), 0, 0, vals, cols, steps, nslices, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)nslices * 64 * steps_slots * 12.0;
  printf("%-28s %8.1f us  %6.0f GB/s\n", name, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
  const size_t total_slots = 500u * 1000 * 1000;  // 6 GB of (8 + 4)-byte slots, like the velocity matrix
  double* vals; int* cols; double* out;
  CK(hipMalloc(&vals, total_slots * 8)); CK(hipMalloc(&cols, total_slots * 4)); CK(hipMalloc(&out, 8));
  CK(hipMemset(vals, 0, total_slots * 8)); CK(hipMemset(cols, 0, total_slots * 4));
  run<2, 2, 4>("16B val + 8B col, unroll 4", vals, cols, out, total_slots);
  run<2, 2, 8>("16B val + 8B col, unroll 8", vals, cols, out, total_slots);
  run<4, 4, 2>("32B val + 16B col, unroll 2", vals, cols, out, total_slots);
  run<4, 4, 4>("32B val + 16B col, unroll 4", vals, cols, out, total_slots);
  run<8, 8, 2>("64B val + 32B col, unroll 2", vals, cols, out, total_slots);
  run<2, 2, 4>("16B val + 8B col, unroll 4", vals, cols, out, total_slots);
  return 0;
}
