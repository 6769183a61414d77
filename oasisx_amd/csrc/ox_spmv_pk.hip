// Packed-stream SpMV (ox_sell::pk_*): the lossless 16-bit column codes and 1-byte value codes of
// ox_spmv.hip regrouped so that one lane reads 8 consecutive entries of its row with ONE 16-byte
// and ONE 8-byte load.  Why: with 2 + 1 bytes per entry the 2-entries-per-load layout of the f64
// stream leaves the kernel bound by the number of vector-memory instructions the CU's address
// unit can take (4 per 384 matrix bytes), not by HBM: the r01 pressure SpMV moved its 139 MB at
// 3.6-4.0 TB/s whatever was done to latency.  Packed: 2.25 + 8 gathers per 8 entries instead of 16.
//
// Optional x window in LDS (k_spmv_pk<..., WIN = true>): a block owns `spb` consecutive slices;
// when the columns its rows touch span at most the LDS budget (P1 spaces in the tiled row order:
// rows +- one tile plane) the block stages x[lo .. lo+n) once with 16-byte loads and the per-entry
// gathers become ds_read_b64.  The span of every block is measured at set-up (k_pk_windows); a
// block whose span does not fit gathers from global memory as before -- speed only, never
// correctness, and the order of the sums is that of k_spmv: results are bit-identical.
#include "ox_common.h"
#include "ox_kernels.h"
#include <stdlib.h>

typedef unsigned short pk_v2h __attribute__((ext_vector_type(2)));
typedef unsigned int pk_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int pk_v2u __attribute__((ext_vector_type(2)));
typedef double pk_v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int pk_wave_min(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int pk_wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- set-up -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pk_count(ox_sell A, int64_t *__restrict__ cnt) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= A.n_slices) return;
  const int npair = (int)((A.slice_ptr[s + 1] - A.slice_ptr[s]) >> 7);
  cnt[s + 1] = (npair + 3) >> 2;
  if (s == 0) cnt[0] = 0;
}

// in-place inclusive scan of v[1..n] (v[0] = 0): one block, set-up time only
__global__ __launch_bounds__(1024) void k_pk_scan(int64_t *v, int n) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int a = 1 + t * per, b = min(n + 1, a + per);
  int64_t s = 0;
  for (int i = a; i < b; ++i) s += v[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int64_t add = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int64_t run = t ? part[t - 1] : 0;
  for (int i = a; i < b; ++i) {
    run += v[i];
    v[i] = run;
  }
}

extern "C" int ox_sell_pack_plan(const ox_sell *A, int64_t *pk_ptr, int64_t *n_groups, void *stream) {
  if (!A || !pk_ptr || !n_groups) OX_FAIL("ox_sell_pack_plan: null argument");
  hipStream_t st = ox_stream(stream);
  *n_groups = 0;
  if (A->n_slices == 0) {
    OX_HIP(hipMemsetAsync(pk_ptr, 0, sizeof(int64_t), st));
    return 0;
  }
  hipLaunchKernelGGL(k_pk_count, dim3((A->n_slices + 255) / 256), dim3(256), 0, st, *A, pk_ptr);
  OX_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pk_scan, dim3(1), dim3(1024), 0, st, pk_ptr, A->n_slices);
  OX_LAUNCH_CHECK();
  OX_HIP(hipMemcpyAsync(n_groups, pk_ptr + A->n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  return 0;
}

// Header of a group (16 ints, read by lanes 0..15 of the wave with ONE vector load that is
// prefetched together with the codes -- no scalar memory load sits on the wave's critical path):
//   [0..7] the two bases (lo, hi) of the group's 4 entry pairs
//   [8]  1: this slice keeps its int32 columns (set in every group of the slice)
//   [9]  groups of the slice      [10] index of the group in its slice     [11] entry pairs of the slice
//   [12],[13] slice_ptr[slice] (low, high word): where the slice's f64 values / int32 columns start
//   [14] the slice
#define PK_HDR 16
// one wave per slice: the dual-base encoding of k_compress_cols (ox_spmv.hip), per entry pair
__global__ __launch_bounds__(256) void k_pk_pack(ox_sell A, uint16_t *__restrict__ pk_cols,
                                                 int32_t *__restrict__ pk_base, uint8_t *__restrict__ pk_vals,
                                                 const uint8_t *__restrict__ vcode, int zero_code,
                                                 unsigned long long *n_fallback) {
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= A.n_slices) return;
  const int64_t base = A.slice_ptr[slice];
  const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);
  const int64_t g0 = A.pk_ptr[slice];
  const int ng = (int)(A.pk_ptr[slice + 1] - g0);
  const int2 *cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
  const int64_t row = (int64_t)slice * 64 + lane;
  const int64_t nlim = A.n_rows < A.n_cols ? A.n_rows : A.n_cols;
  const int own = row < nlim ? (int)row : 0;  // padding rule of build_sell: own row, value 0
  const int BIG = 0x7fffffff;
  bool ok = true;
  for (int p = 0; p < ng * 4; ++p) {
    const int2 c = p < npair ? cp[(size_t)p * 64] : make_int2(own, own);
    const int lo = pk_wave_min(min(c.x, c.y));
    const bool fx = c.x - lo >= 32768, fy = c.y - lo >= 32768;
    const int lo2 = pk_wave_min(min(fx ? c.x : BIG, fy ? c.y : BIG));
    const int hi2 = pk_wave_max(max(fx ? c.x : -1, fy ? c.y : -1));
    const bool fits = (lo2 == BIG) || (hi2 - lo2 < 32768);
    ok = ok && fits;
    const size_t e = ((size_t)(g0 + (p >> 2)) * 64 + lane) * 8 + (p & 3) * 2;
    pk_cols[e] = !fits ? 0 : (unsigned short)(fx ? (0x8000 | (c.x - lo2)) : (c.x - lo));
    pk_cols[e + 1] = !fits ? 0 : (unsigned short)(fy ? (0x8000 | (c.y - lo2)) : (c.y - lo));
    if (lane == 0) {
      int32_t *h = pk_base + (size_t)(g0 + (p >> 2)) * PK_HDR;
      h[(p & 3) * 2] = lo;
      h[(p & 3) * 2 + 1] = lo2 == BIG ? lo : lo2;
    }
    if (pk_vals) {
      const uint8_t *vc = vcode + base + (size_t)p * 128 + lane * 2;
      pk_vals[e] = p < npair ? vc[0] : (uint8_t)zero_code;
      pk_vals[e + 1] = p < npair ? vc[1] : (uint8_t)zero_code;
    }
  }
  ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
  if (lane < ng) {
    int32_t *h = pk_base + (size_t)(g0 + lane) * PK_HDR;
    h[8] = ok ? 0 : 1;
    h[9] = ng;
    h[10] = lane;
    h[11] = npair;
    h[12] = (int32_t)(base & 0xffffffffll);
    h[13] = (int32_t)(base >> 32);
    h[14] = slice;
    h[15] = 0;
  }
  if (lane == 0 && ng > 0 && !ok && n_fallback) atomicAdd(n_fallback, 1ull);
}

extern "C" int ox_sell_pack(const ox_sell *A, uint16_t *pk_cols, int32_t *pk_base, uint8_t *pk_vals,
                            const uint8_t *vcode, int zero_code, int64_t *n_fallback, void *stream) {
  if (!A || !A->pk_ptr || !pk_cols || !pk_base) OX_FAIL("ox_sell_pack: null argument");
  if (pk_vals && !vcode) OX_FAIL("ox_sell_pack: pk_vals without vcode");
  hipStream_t st = ox_stream(stream);
  if (n_fallback) *n_fallback = 0;
  if (A->n_slices == 0) return 0;
  unsigned long long *cnt = nullptr;
  if (n_fallback) {
    OX_HIP(hipMalloc(&cnt, sizeof(*cnt)));
    OX_HIP(hipMemsetAsync(cnt, 0, sizeof(*cnt), st));
  }
  hipLaunchKernelGGL(k_pk_pack, dim3((A->n_slices + 3) / 4), dim3(256), 0, st, *A, pk_cols, pk_base, pk_vals,
                     vcode, zero_code, cnt);
  OX_LAUNCH_CHECK();
  if (n_fallback) {
    unsigned long long h = 0;
    OX_HIP(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    OX_HIP(hipStreamSynchronize(st));
    OX_HIP(hipFree(cnt));
    *n_fallback = (int64_t)h;
  }
  return 0;
}

// ---- the kernel ---------------------------------------------------------------------------
// blockDim.x / 64 waves per block; a wave owns a run of consecutive slices, whose groups are
// consecutive in memory: it streams ONE contiguous run of groups and always has the next two groups
// (codes + header) in flight, across slice boundaries and across the window barrier.  Everything
// the loop needs to know about a group comes with the group's header through the vector pipeline.
// DICT: values from pk_vals + LDS dictionary, else f64 from A.vals (matrices that change: A);
// WIN: x window in LDS (NC = 1).  DIAG (tools only): every gather reads the row's own x.
template <int NC, int EPI, bool DICT, bool WIN, bool DIAG>
__global__ __launch_bounds__(1024) void k_spmv_pk(ox_sell A, const double *__restrict__ x,
                                                  double *__restrict__ y, const double *__restrict__ dinv,
                                                  const double *__restrict__ aux, double *__restrict__ partial,
                                                  const int *__restrict__ done_flag, int spw, int wmax) {
  constexpr int NV = (EPI == OX_EPI_NONE) ? 1 : (EPI == OX_EPI_BCGS_T ? 2 * NC : NC);
  __shared__ double red[16 * NV];
  __shared__ double dict[DICT ? 256 : 1];
  extern __shared__ double xw[];  // WIN: the block's x window
  if (done_flag && *done_flag) return;
  const int nwaves = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = ox_xcd_remap(blockIdx.x, gridDim.x);
  // spw > 0: `spw` slices per wave; spw <= 0: the slices are dealt evenly over all waves of the
  // grid (a persistent grid of n_CU x k blocks: every CU gets the same work to within one slice)
  int s_begin, s_end, sb_begin, sb_end;
  if (spw > 0) {
    s_begin = min((b * nwaves + wave) * spw, A.n_slices);
    s_end = min(s_begin + spw, A.n_slices);
    sb_begin = min(b * nwaves * spw, A.n_slices);
    sb_end = min(sb_begin + nwaves * spw, A.n_slices);
  } else {
    const int64_t W = (int64_t)gridDim.x * nwaves, w = (int64_t)b * nwaves + wave;
    s_begin = (int)(w * A.n_slices / W);
    s_end = (int)((w + 1) * A.n_slices / W);
    sb_begin = (int)((int64_t)b * nwaves * A.n_slices / W);
    sb_end = (int)((int64_t)(b + 1) * nwaves * A.n_slices / W);
  }
  s_begin = __builtin_amdgcn_readfirstlane(s_begin);
  s_end = __builtin_amdgcn_readfirstlane(s_end);
  const int64_t G0 = A.pk_ptr[s_begin], G1 = A.pk_ptr[s_end];
  const pk_v4u *__restrict__ cq = reinterpret_cast<const pk_v4u *>(A.pk_cols) + lane;
  const pk_v2u *__restrict__ vq = DICT ? reinterpret_cast<const pk_v2u *>(A.pk_vals) + lane : nullptr;
  const int *__restrict__ hq = A.pk_base + (lane & (PK_HDR - 1));
  struct Grp {
    pk_v4u c;
    pk_v2u v;
    int h;
  };
  auto load_group = [&](int64_t g) {
    Grp r;
    r.c = __builtin_nontemporal_load(cq + g * 64);
    r.v = DICT ? __builtin_nontemporal_load(vq + g * 64) : pk_v2u{0u, 0u};
    r.h = hq[g * PK_HDR];  // lanes 16..63 repeat lanes 0..15: same 64 bytes, one request
    return r;
  };
  // the wave's first two groups are in flight before the block meets at the window barrier
  Grp cur{}, nxt{};
  if (G0 < G1) cur = load_group(G0);
  if (G0 + 1 < G1) nxt = load_group(G0 + 1);
  int wlo = 0, wn = 0;
  if (DICT) {
    if ((int)threadIdx.x < A.n_dict) dict[threadIdx.x] = A.vdict[threadIdx.x];
  }
  if (WIN) {
    // the window is centred on the block's own rows: in the tiled row order of a P1 space the columns
    // of a row lie within +- one tile plane of it; whatever falls outside (neighbours across a tile
    // edge, ghost columns) is gathered from global memory, group by group
    const int64_t r0 = (int64_t)sb_begin * 64, r1 = min((int64_t)sb_end * 64, A.n_cols);
    int64_t lo = r0 - (wmax - (r1 - r0)) / 2;
    lo = lo < 0 ? 0 : lo & ~(int64_t)1;
    wlo = (int)lo;
    wn = (int)min((int64_t)wmax, A.n_cols - lo);
    if (wn < 0) wn = 0;
    // 16-byte loads (wlo is even), 4 in flight per thread: coalesced, every line once
    const pk_v2d *src = reinterpret_cast<const pk_v2d *>(x + wlo);
    const int n2 = wn >> 1, T = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n2; i0 += 4 * T) {
      pk_v2d t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * T < n2) t[u] = src[i0 + u * T];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * T < n2) {
          xw[2 * (i0 + u * T)] = t[u].x;
          xw[2 * (i0 + u * T) + 1] = t[u].y;
        }
    }
    if ((wn & 1) && threadIdx.x == 0) xw[wn - 1] = x[wlo + wn - 1];
  }
  if (DICT || WIN) __syncthreads();
  double s[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) s[i] = 0.0;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  int slice = s_begin;
  auto finish_slice = [&]() {  // the slice's rows are complete: epilogue, store, dot products
    const int64_t row = (int64_t)slice * 64 + lane;
    if (row < A.n_rows) {
      if (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T) {
        const double d = dinv[row];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] *= d;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) y[row * NC + c] = acc[c];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (EPI == OX_EPI_DOT) s[c] = fma(x[row * NC + c], acc[c], s[c]);
        if (EPI == OX_EPI_BCGS_V) s[c] = fma(aux[row * NC + c], acc[c], s[c]);
        if (EPI == OX_EPI_BCGS_T) {
          s[c] = fma(acc[c], acc[c], s[c]);
          s[NC + c] = fma(acc[c], x[row * NC + c], s[NC + c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    ++slice;
  };
  int64_t g = G0;
  while (g < G1) {
    const int fb = __builtin_amdgcn_readlane(cur.h, 8), ngs = __builtin_amdgcn_readlane(cur.h, 9);
    const int gi = __builtin_amdgcn_readlane(cur.h, 10), npair = __builtin_amdgcn_readlane(cur.h, 11);
    const int64_t sbase = ((int64_t)__builtin_amdgcn_readlane(cur.h, 13) << 32) |
                          (unsigned)__builtin_amdgcn_readlane(cur.h, 12);
    if (fb) {  // (gi == 0 here) the whole slice from its int32 columns and f64 values
      const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + sbase) + lane;
      const double2 *__restrict__ vp = reinterpret_cast<const double2 *>(A.vals + sbase) + lane;
      for (int k = 0; k < npair; ++k) {
        const int2 c = cp[(size_t)k * 64];
        const double2 v = vp[(size_t)k * 64];
#pragma unroll
        for (int c2 = 0; c2 < NC; ++c2) acc[c2] = fma(v.x, x[(size_t)c.x * NC + c2], acc[c2]);
#pragma unroll
        for (int c2 = 0; c2 < NC; ++c2) acc[c2] = fma(v.y, x[(size_t)c.y * NC + c2], acc[c2]);
      }
      finish_slice();
      g += ngs;  // the stream resumes at the next slice's first group
      if (g >= G1) break;
      cur = load_group(g);
      nxt = load_group(g + 1 < G1 ? g + 1 : G1 - 1);
      continue;
    }
    const unsigned cw[4] = {cur.c.x, cur.c.y, cur.c.z, cur.c.w};
    int col[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bl = __builtin_amdgcn_readlane(cur.h, 2 * j), bh = __builtin_amdgcn_readlane(cur.h, 2 * j + 1);
      const int d0 = cw[j] & 0xffff, d1 = cw[j] >> 16;
      col[2 * j] = ((d0 & 0x8000) ? bh : bl) + (d0 & 0x7fff);
      col[2 * j + 1] = ((d1 & 0x8000) ? bh : bl) + (d1 & 0x7fff);
    }
    double v[8];
    if (DICT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = dict[(cur.v.x >> (8 * j)) & 0xff];
        v[4 + j] = dict[(cur.v.y >> (8 * j)) & 0xff];
      }
    } else {
      const pk_v2d *__restrict__ vp = reinterpret_cast<const pk_v2d *>(A.vals + sbase) + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pk_v2d vv = {0.0, 0.0};
        if (4 * gi + j < npair) vv = __builtin_nontemporal_load(vp + (size_t)(4 * gi + j) * 64);
        v[2 * j] = vv.x;
        v[2 * j + 1] = vv.y;
      }
    }
    double xv[8][NC];
    bool lds_ok = false;
    if (WIN && !DIAG) {  // one wave-uniform decision per group: every column of every lane inside?
      bool in = true;
#pragma unroll
      for (int j = 0; j < 8; ++j) in = in && ((unsigned)(col[j] - wlo) < (unsigned)wn);
      lds_ok = __builtin_amdgcn_ballot_w64(!in) == 0;
    }
    if (DIAG) {
      const int64_t row = (int64_t)slice * 64 + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c)  // col >> 30 == 0: the decode stays live, the address is the row's own
          xv[j][c] = x[(size_t)((row < A.n_rows ? row : 0) + (col[j] >> 30)) * NC + c];
    } else if (WIN && lds_ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j][0] = xw[col[j] - wlo];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) xv[j][c] = x[(size_t)col[j] * NC + c];
    }
    // two groups ahead in this wave's run -- issued AFTER the gathers: vector loads return in order,
    // so the wait for the gathers below leaves these three in flight
    // (unconditional -- the last groups are simply loaded again -- so that the compiler's wait
    // counts stay exact across the branch)
    const Grp nn = load_group(g + 2 < G1 ? g + 2 : G1 - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = fma(v[j], xv[j][c], acc[c]);
    if (gi == ngs - 1) finish_slice();
    cur = nxt;
    nxt = nn;
    ++g;
  }
  if (EPI != OX_EPI_NONE) {
    // fixed-order block sum over up to 16 waves
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double t = ox_wave_sum(s[i]);
      if (lane == 0) red[i * 16 + wave] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int w = 0; w < nwaves; ++w) t += red[i * 16 + w];
        partial[(size_t)blockIdx.x * NV + i] = t;
      }
    }
  }
}

// ---- the cyclic persistent kernel -------------------------------------------------------------
// Grid = n_CU x k blocks, all resident.  The blocks that share an XCD (equal blockIdx % 8) own one
// contiguous eighth of the slices (x gathers stay in that XCD's L2); inside it the XCD's waves take
// the slices round-robin: wave i takes slices i, i + W, i + 2W, ... -- at any moment the chip works
// on ONE contiguous window of the matrix (DRAM pages stay open, as in a plain streaming kernel),
// yet every wave lives for the whole launch and keeps a ring of three groups (codes + header) in
// flight, ~3 KB per wave, independent of how short a slice is.  (The r01 kernel, one slice per
// wave, and a contiguous-range persistent variant both sat at 3.5-4.4 TB/s: one group in flight per
// wave and ~1.5 us of loaded latency per dependent round is exactly that rate.)
// The ring is unrolled (no register rotation: copying a register with a load in flight forces a
// full wait) and every load of the loop is unconditional, so the compiler's wait counts are exact.
template <int NC, int EPI, bool DICT>
__global__ __launch_bounds__(1024) void k_spmv_pkc(ox_sell A, const double *__restrict__ x,
                                                   double *__restrict__ y, const double *__restrict__ dinv,
                                                   const double *__restrict__ aux, double *__restrict__ partial,
                                                   const int *__restrict__ done_flag) {
  constexpr int NV = (EPI == OX_EPI_NONE) ? 1 : (EPI == OX_EPI_BCGS_T ? 2 * NC : NC);
  __shared__ double red[16 * NV];
  __shared__ double dict[DICT ? 256 : 1];
  if (done_flag && *done_flag) return;
  const int nwaves = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7;
  const int chunk = (A.n_slices + 7) >> 3;
  const int s_lo = xcd * chunk, s_hi = min(A.n_slices, s_lo + chunk);
  const int stride = (gridDim.x >> 3) * nwaves;  // waves of this XCD
  const int s_first = __builtin_amdgcn_readfirstlane(s_lo + (int)(blockIdx.x >> 3) * nwaves + wave);
  const int64_t g_last = A.pk_ptr[A.n_slices] - 1;  // clamp of the run-out prefetches
  const pk_v4u *__restrict__ cq = reinterpret_cast<const pk_v4u *>(A.pk_cols) + lane;
  const pk_v2u *__restrict__ vq = DICT ? reinterpret_cast<const pk_v2u *>(A.pk_vals) + lane : nullptr;
  const int *__restrict__ hq = A.pk_base + (lane & (PK_HDR - 1));
  struct Grp {
    pk_v4u c;
    pk_v2u v;
    int h;
  };
  auto load_group = [&](int64_t g) {
    Grp r;
    r.c = __builtin_nontemporal_load(cq + g * 64);
    r.v = DICT ? __builtin_nontemporal_load(vq + g * 64) : pk_v2u{0u, 0u};
    r.h = hq[g * PK_HDR];  // lanes 16..63 repeat lanes 0..15: same 64 bytes, one request
    return r;
  };
  // prefetch iterator: runs three groups ahead of the consumer through the same visit order.  The
  // group ranges of the wave's slices come from a per-wave table -- lane l holds pk_ptr[s], pk_ptr[s+1]
  // of the wave's l-th slice, ONE vector load per 64 slices -- read with v_readlane: no memory
  // access of the loop depends on a scalar load issued inside it.
  const int n_mine = s_first < s_hi ? (s_hi - s_first + stride - 1) / stride : 0;  // slices of this wave
  int64_t tab0 = 0, tab1 = 0;
  auto load_table = [&](int k0) {
    const int64_t sl = (int64_t)s_first + (int64_t)(k0 + lane) * stride;
    if (k0 + lane < n_mine) {
      tab0 = A.pk_ptr[sl];
      tab1 = A.pk_ptr[sl + 1];
    }
  };
  load_table(0);
  int pk = 0;                    // index (among the wave's slices) of the iterator's slice
  int64_t pg = 0, pg_end = 0;    // its next group / end
  auto table_entry = [&](int k) {
    const int l = k & 63;
    pg = ((int64_t)__builtin_amdgcn_readlane((int)(tab0 >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tab0, l);
    pg_end = ((int64_t)__builtin_amdgcn_readlane((int)(tab1 >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tab1, l);
  };
  if (n_mine > 0) table_entry(0);
  auto next_group = [&]() -> int64_t {
    if (pg >= pg_end && pk < n_mine) {  // (a slice always has >= 1 group)
      ++pk;
      if (pk < n_mine) {
        if ((pk & 63) == 0) load_table(pk);
        table_entry(pk);
      }
    }
    if (pk >= n_mine) return g_last < 0 ? 0 : g_last;
    return pg++;
  };
  if (DICT) {
    if ((int)threadIdx.x < A.n_dict) dict[threadIdx.x] = A.vdict[threadIdx.x];
    __syncthreads();
  }
  double s[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) s[i] = 0.0;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  const char *__restrict__ xb = reinterpret_cast<const char *>(x);
  bool live = s_first < s_hi && g_last >= 0;
  // consume one ring slot, then refill it; returns false after the wave's last slice
  auto consume = [&](Grp &R) {
    const int fb = __builtin_amdgcn_readlane(R.h, 8), ngs = __builtin_amdgcn_readlane(R.h, 9);
    const int gi = __builtin_amdgcn_readlane(R.h, 10), npair = __builtin_amdgcn_readlane(R.h, 11);
    const int slice = __builtin_amdgcn_readlane(R.h, 14);
    const int64_t sbase = ((int64_t)__builtin_amdgcn_readlane(R.h, 13) << 32) |
                          (unsigned)__builtin_amdgcn_readlane(R.h, 12);
    const int64_t row = (int64_t)slice * 64 + lane;
    const int64_t gn = next_group();  // what this slot is refilled with (scalar work, ahead of the gathers)
    double v[8], xv[8][NC];
    if (fb) {  // this slice keeps its int32 columns: all of it with its first group, nothing with the others
      if (gi == 0) {
        const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + sbase) + lane;
        const double2 *__restrict__ vp = reinterpret_cast<const double2 *>(A.vals + sbase) + lane;
        for (int k = 0; k < npair; ++k) {
          const int2 c = cp[(size_t)k * 64];
          const double2 vv = vp[(size_t)k * 64];
#pragma unroll
          for (int c2 = 0; c2 < NC; ++c2) acc[c2] = fma(vv.x, x[(size_t)c.x * NC + c2], acc[c2]);
#pragma unroll
          for (int c2 = 0; c2 < NC; ++c2) acc[c2] = fma(vv.y, x[(size_t)c.y * NC + c2], acc[c2]);
        }
      }
      R = load_group(gn);
    } else {
      const unsigned cw[4] = {R.c.x, R.c.y, R.c.z, R.c.w};
      unsigned off[8];  // byte offsets into x: 32 bits (the launcher checks n_cols * NC * 8 < 4 GiB)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int bl = __builtin_amdgcn_readlane(R.h, 2 * j), bh = __builtin_amdgcn_readlane(R.h, 2 * j + 1);
        const int d0 = cw[j] & 0xffff, d1 = cw[j] >> 16;
        off[2 * j] = (unsigned)(((d0 & 0x8000) ? bh : bl) + (d0 & 0x7fff)) * (unsigned)(8 * NC);
        off[2 * j + 1] = (unsigned)(((d1 & 0x8000) ? bh : bl) + (d1 & 0x7fff)) * (unsigned)(8 * NC);
      }
      if (DICT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = dict[(R.v.x >> (8 * j)) & 0xff];
          v[4 + j] = dict[(R.v.y >> (8 * j)) & 0xff];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) xv[j][c] = *reinterpret_cast<const double *>(xb + off[j] + 8 * c);
      if (!DICT) {
        const pk_v2d *__restrict__ vp = reinterpret_cast<const pk_v2d *>(A.vals + sbase) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = min(4 * gi + j, npair - 1);  // unconditional load; pairs past the row's end count 0
          const pk_v2d vv = __builtin_nontemporal_load(vp + (size_t)k * 64);
          v[2 * j] = 4 * gi + j < npair ? vv.x : 0.0;
          v[2 * j + 1] = 4 * gi + j < npair ? vv.y : 0.0;
        }
      }
      // refill AFTER the gathers: vector loads return in order, so the wait for the gathers leaves the
      // ring's loads in flight
      __builtin_amdgcn_sched_barrier(0);
      R = load_group(gn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fma(v[j], xv[j][c], acc[c]);
    }
    if (gi == ngs - 1) {  // the slice's rows are complete: epilogue, store, dot products
      if (row < A.n_rows) {
        if (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T) {
          const double d = dinv[row];
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] *= d;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) y[row * NC + c] = acc[c];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (EPI == OX_EPI_DOT) s[c] = fma(x[row * NC + c], acc[c], s[c]);
          if (EPI == OX_EPI_BCGS_V) s[c] = fma(aux[row * NC + c], acc[c], s[c]);
          if (EPI == OX_EPI_BCGS_T) {
            s[c] = fma(acc[c], acc[c], s[c]);
            s[NC + c] = fma(acc[c], x[row * NC + c], s[NC + c]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = 0.0;
      live = slice + stride < s_hi;
    }
  };
  if (live) {
    Grp r0 = load_group(next_group()), r1 = load_group(next_group()), r2 = load_group(next_group());
    while (true) {
      consume(r0);
      if (!live) break;
      consume(r1);
      if (!live) break;
      consume(r2);
      if (!live) break;
    }
  }
  if (EPI != OX_EPI_NONE) {
    // fixed-order block sum over up to 16 waves
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double t = ox_wave_sum(s[i]);
      if (lane == 0) red[i * 16 + wave] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int w = 0; w < nwaves; ++w) t += red[i * 16 + w];
        partial[(size_t)blockIdx.x * NV + i] = t;
      }
    }
  }
}

// Launch configuration of the packed kernel for one matrix, cached per pattern (pk_ptr).
struct PkPlan {
  const void *key;  // pk_ptr
  int n_slices;
  int waves, spw, nblk;
  int wmax;         // LDS doubles of the window kernel (0: windows off)
  int n_fit;        // blocks whose window fits
};
#include <vector>
static std::vector<PkPlan> g_pk_plans;
static int g_pk_mode = -1;  // 0 off, 1 packed stream, 2 packed + LDS x window (NC = 1), 9 diag
static int g_pk_waves = 16, g_pk_spw = 4, g_pk_wmax = 12800;

static void pk_env() {
  if (g_pk_mode >= 0) return;
  const char *e = getenv("OX_PK_MODE");
  g_pk_mode = e ? atoi(e) : 2;
  if ((e = getenv("OX_PK_WAVES"))) g_pk_waves = atoi(e);
  if ((e = getenv("OX_PK_SPW"))) g_pk_spw = atoi(e);
  if ((e = getenv("OX_PK_WMAX"))) g_pk_wmax = atoi(e);
  if (g_pk_waves < 1 || g_pk_waves > 16) g_pk_waves = 16;
}
extern "C" int ox_set_pk_mode(int mode, int waves, int spw, int wmax) {
  pk_env();
  g_pk_mode = mode;
  if (waves >= 1 && waves <= 16) g_pk_waves = waves;
  if (spw != 0) g_pk_spw = spw;
  if (wmax >= 0) g_pk_wmax = wmax;
  g_pk_plans.clear();
  return 0;
}
int ox_pk_mode() {
  pk_env();
  return g_pk_mode;
}

static int g_pk_cus = 0;
int ox_spmv_pk_blocks(const ox_sell *A) {
  pk_env();
  if (g_pk_spw > 0 && g_pk_mode != 3) {
    const int spb = g_pk_waves * g_pk_spw;
    return (A->n_slices + spb - 1) / spb;
  }
  if (!g_pk_cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) g_pk_cus = pr.multiProcessorCount;
    if (g_pk_cus < 1) g_pk_cus = 256;
  }
  const int want = g_pk_cus * (-g_pk_spw > 0 ? -g_pk_spw : 1);  // spw = -k: k blocks per CU
  const int most = (A->n_slices + g_pk_waves - 1) / g_pk_waves;  // at least one slice per wave
  int n = want < most ? want : (most > 0 ? most : 0);
  if (g_pk_mode == 3) n = n > 8 ? (n + 7) & ~7 : (n > 0 ? 8 : 0);  // the cyclic kernel deals whole XCD groups
  return n;
}

static int pk_plan(const ox_sell *A, int ncomp, hipStream_t st, PkPlan **out) {
  for (auto &p : g_pk_plans)
    if (p.key == A->pk_ptr && p.n_slices == A->n_slices) {
      *out = &p;
      return 0;
    }
  PkPlan p{};
  p.key = A->pk_ptr;
  p.n_slices = A->n_slices;
  p.waves = g_pk_waves;
  p.spw = g_pk_spw;
  p.nblk = ox_spmv_pk_blocks(A);
  p.wmax = (g_pk_mode == 2 && ncomp == 1) ? g_pk_wmax : 0;
  g_pk_plans.push_back(p);
  *out = &g_pk_plans.back();
  return 0;
}

extern "C" int ox_pk_plan_info(const ox_sell *A, int *nblk, int *n_fit, int *wmax) {
  PkPlan *p = nullptr;
  pk_env();
  if (!A || !A->pk_ptr) OX_FAIL("ox_pk_plan_info: matrix has no packed stream");
  if (pk_plan(A, 1, nullptr, &p)) return -1;
  if (nblk) *nblk = p->nblk;
  if (n_fit) *n_fit = p->n_fit;
  if (wmax) *wmax = p->wmax;
  return 0;
}

// number of per-block partials the packed launch of A writes (Krylov workspace sizing / reduction)
template <int NC, int EPI>
static int pk_launch_t(const ox_sell *A, const double *x, double *y, const double *dinv, const double *aux,
                       double *partial, const int *done, hipStream_t st, const PkPlan &p) {
  const bool dict = A->pk_vals && A->vdict && A->n_dict >= 1 && A->n_dict <= 256;
  const bool win = NC == 1 && p.wmax > 0;
  const dim3 grid(p.nblk), block(p.waves * 64);
  const size_t lds = win ? (size_t)(p.wmax + 2) * sizeof(double) : 0;
#define OX_PK_GO(D, W, G)                                                                             \
  do {                                                                                                \
    auto kern = k_spmv_pk<NC, EPI, D, W, G>;                                                          \
    if (lds > 48 * 1024)                                                                              \
      OX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
    hipLaunchKernelGGL(kern, grid, block, lds, st, *A, x, y, dinv, aux, partial, done, p.spw, p.wmax);  \
  } while (0)
  if (g_pk_mode == 3) {
    if (dict) hipLaunchKernelGGL((k_spmv_pkc<NC, EPI, true>), grid, block, 0, st, *A, x, y, dinv, aux, partial, done);
    else hipLaunchKernelGGL((k_spmv_pkc<NC, EPI, false>), grid, block, 0, st, *A, x, y, dinv, aux, partial, done);
  } else if (g_pk_mode == 9) {
    if (dict) OX_PK_GO(true, false, true);
    else OX_PK_GO(false, false, true);
  } else if (NC == 1 && win) {
    if constexpr (NC == 1) {
      if (dict) OX_PK_GO(true, true, false);
      else OX_PK_GO(false, true, false);
    }
  } else {
    if (dict) OX_PK_GO(true, false, false);
    else OX_PK_GO(false, false, false);
  }
#undef OX_PK_GO
  OX_LAUNCH_CHECK();
  return 0;
}

// returns 1 when the matrix has no packed stream (caller falls back to k_spmv)
int ox_spmv_pk_launch(const ox_sell *A, const double *x, double *y, int ncomp, int epi, const double *dinv,
                      const double *aux, double *partial, const int *done, hipStream_t st) {
  pk_env();
  if (g_pk_mode == 0 || !A->pk_ptr || !A->pk_cols || !A->pk_base) return 1;
  PkPlan *p = nullptr;
  if (pk_plan(A, ncomp, st, &p)) return -1;
  if (p->nblk == 0) return 0;
  if (ox_prof_on) ox_prof_start(OX_TAG_SPMV(ncomp, epi), st, A->n_rows);
  int rc = -1;
#define OX_PK_CASE(NC, E) \
  if (ncomp == NC && epi == E) rc = pk_launch_t<NC, E>(A, x, y, dinv, aux, partial, done, st, *p);
#define OX_PK_NC(NC) \
  OX_PK_CASE(NC, OX_EPI_NONE) OX_PK_CASE(NC, OX_EPI_DOT) OX_PK_CASE(NC, OX_EPI_BCGS_V) OX_PK_CASE(NC, OX_EPI_BCGS_T)
  OX_PK_NC(1) OX_PK_NC(2) OX_PK_NC(3)
#undef OX_PK_NC
#undef OX_PK_CASE
  if (ox_prof_on) ox_prof_stop(st);
  return rc;
}
