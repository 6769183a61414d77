// SELL-64 SpMV / multi-vector SpMM, BLAS-1, set_bc, zero-rows, Jacobi setup.
// Memory-bound float64 kernels for gfx950: lane = row, 16-B value loads, XCD-aware
// slice mapping.  No MFMA: the path is sparse and HBM-bound.
#include "ox_common.h"
#include "ox_kernels.h"
#include <stdlib.h>
#include <type_traits>

thread_local char ox_err_buf[512] = "";

extern "C" int ox_version(void) { return 100; }
extern "C" const char *ox_last_error(void) { return ox_err_buf; }
extern "C" int ox_sell_kv(void) { return OX_KV; }

extern "C" int ox_device_info(int *n_cu, char *name, int name_len) {
  int dev = 0;
  OX_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  OX_HIP(hipGetDeviceProperties(&p, dev));
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// SpMV.  One wave per slice (lane = row), 4 slices per 256-thread block.  Per k-pair a lane
// loads 16 B of values + 8 B of columns (coalesced over the wave: 1 KiB + 512 B) and gathers
// NC doubles of x per entry.  EPI selects the fused epilogue the Krylov loops need.
// ---------------------------------------------------------------------------------------
template <int NC, int EPI, int VAR>
__global__ __launch_bounds__(256) void k_spmv(ox_sell A, const double *__restrict__ x,
                                              double *__restrict__ y,
                                              const double *__restrict__ dinv,
                                              const double *__restrict__ aux,
                                              double *__restrict__ partial,
                                              const int *__restrict__ done_flag,
                                              const int32_t *__restrict__ slice_list, int n_list, OxEpiDinv ED) {
  constexpr int NV = (EPI == OX_EPI_NONE) ? 1 : ox_epi_nv(EPI, NC);
  __shared__ double red[4 * NV];
  __shared__ double dict[(VAR & 4) ? 256 : 1];
  if (done_flag && *done_flag) return;
  if (VAR & 4) {  // value dictionary (<= 256 distinct values in the whole matrix): 1 B per entry
    if ((int)threadIdx.x < A.n_dict) dict[threadIdx.x] = A.vdict[threadIdx.x];
    __syncthreads();
  }
  // Persistent grid (<= OX_SPMV_MAX_BLOCKS blocks, a multiple of 8): the blocks that share an XCD
  // (equal blockIdx % 8) stride together over ONE contiguous eighth of the slice groups, so
  // at any time an XCD's L2 serves a compact window of rows and of x.
  // slice_list (interior / boundary split of a partitioned operator): the launch covers that list
  const int n_sl = slice_list ? n_list : A.n_slices;
  const int ngroups = (n_sl + 3) >> 2;
  const int per = gridDim.x >> 3;
  const int chunk = (ngroups + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int g_begin = xcd * chunk;
  const int g_end = min(ngroups, g_begin + chunk);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) s[i] = 0.0;
  for (int g = g_begin + (blockIdx.x >> 3); g < g_end; g += per) {
    const int li = g * 4 + wave;
    if (li >= n_sl) continue;
    const int slice = __builtin_amdgcn_readfirstlane(slice_list ? slice_list[li] : li);  // wave-uniform: scalar loads below
    const int64_t row = (int64_t)slice * 64 + lane;
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const int64_t base = A.slice_ptr[slice];
    const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);  // width / OX_KV
    const double2 *__restrict__ vp = reinterpret_cast<const double2 *>(A.vals + base) + lane;
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef unsigned short v2h __attribute__((ext_vector_type(2)));
    const unsigned short *__restrict__ vcp =
        (VAR & 4) ? reinterpret_cast<const unsigned short *>(A.vcode + base) + lane : nullptr;
    auto load_vals = [&](int k) {
      double2 v;
      if (VAR & 4) {
        const unsigned cc = __builtin_nontemporal_load(vcp + (size_t)k * 64);  // two 1-byte codes
        v.x = dict[cc & 0xff];
        v.y = dict[cc >> 8];
      } else if (VAR & 1) {
        const v2d vv = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(vp + (size_t)k * 64));
        v.x = vv.x;
        v.y = vv.y;
      } else {
        v = vp[(size_t)k * 64];
      }
      return v;
    };
    auto mac = [&](double2 v, int c0, int c1) {
      const double *x0 = x + (size_t)c0 * NC;
      const double *x1 = x + (size_t)c1 * NC;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(v.x, x0[cc], acc[cc]);
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(v.y, x1[cc], acc[cc]);
    };
    // 16-bit column stream (ox_sell_compress_cols): col = cbase[pair][bit 15 of the code] + low 15
    // bits, 10 B per entry instead of 12.  cbase[first pair of the slice][0] < 0: this slice keeps
    // its int32 columns.
    const int2 *__restrict__ cb =
        (VAR & 2) ? reinterpret_cast<const int2 *>(A.cbase) + (base >> 7) : nullptr;
    if ((VAR & 2) && cb[0].x >= 0) {
      const v2h *__restrict__ hp = reinterpret_cast<const v2h *>(A.cols16 + base) + lane;
#pragma unroll 4
      for (int k = 0; k < npair; ++k) {
        const double2 v = load_vals(k);
        const v2h d = __builtin_nontemporal_load(hp + (size_t)k * 64);
        const int2 b = cb[k];
        const int dx = d.x, dy = d.y;
        mac(v, ((dx & 0x8000) ? b.y : b.x) + (dx & 0x7fff), ((dy & 0x8000) ? b.y : b.x) + (dy & 0x7fff));
      }
    } else {
      const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
#pragma unroll 4  // unroll 8 / 16 measured neutral to 17 % slower (tools/spmv_bench.py, round 1)
      for (int k = 0; k < npair; ++k) {
        const double2 v = load_vals(k);
        int2 c;
        if (VAR & 1) {
          const v2i ci = __builtin_nontemporal_load(reinterpret_cast<const v2i *>(cp + (size_t)k * 64));
          c.x = ci.x;
          c.y = ci.y;
        } else {
          c = cp[(size_t)k * 64];
        }
        mac(v, c.x, c.y);
      }
    }
    if (row < A.n_rows) {
      if (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T || EPI == OX_EPI_BCGS_T5) {
        const double d = dinv[row];  // one matrix, one diagonal
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
        if (EPI == OX_EPI_BCGS_T5) {
          const double xs = x[row * NC + c], ah = aux[row * NC + c];
          s[c] = fma(acc[c], acc[c], s[c]);
          s[NC + c] = fma(acc[c], xs, s[NC + c]);
          s[2 * NC + c] = fma(ah, xs, s[2 * NC + c]);
          s[3 * NC + c] = fma(ah, acc[c], s[3 * NC + c]);
          s[4 * NC + c] = fma(xs, xs, s[4 * NC + c]);
        }
        if (EPI == OX_EPI_CG_M2) {
          const double dd = ED.code ? ED.dict[ED.code[row]] : dinv[row];
          s[c] = fma(x[row * NC + c], acc[c], s[c]);
          s[NC + c] = fma(acc[c], dd * acc[c], s[NC + c]);
        }
      }
    }
  }
  if (EPI != OX_EPI_NONE) {
    ox_block_sum_256<NV>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) partial[(size_t)blockIdx.x * NV + i] = s[i];
    }
  }
}


// ---------------------------------------------------------------------------------------
// k_spmv_ps: y = A x from the pair-slot stream of a value-dictionary matrix (ox_sell.ps_*).
// What bounds k_spmv<., ., 7>: the address unit takes ~16-20 cycles per vector-memory instruction of a
// wave whatever its width (tools/ubench/dispatch_rate.hip: 16 loads per wave cost 20.5 us on this grid
// as ushort, dword, dwordx2 or dwordx4, from L1, L2 or the Infinity Cache alike), and a 64-row slice of
// the pressure matrix issues 35 of them (16 gathers of 8 B, 8 + 8 code loads, epilogue): 34 us at
// 128^3 however the wave is scheduled; without the gathers 16 us.  So: fewer, wider instructions.
// One slot = two adjacent columns = one 16-B gather (NC = 1); four slots = one 16-B code load.
// The same per-row order of fused multiply-adds as k_spmv: bit-identical results.
// A wave's dependent memory rounds: (A) done flag, ps_ptr, dictionary, epilogue operands;
// (B) codes + bases of the first two groups; (C) gathers; dictionary reads from a wave-private LDS
// copy (no block barrier).
// ---------------------------------------------------------------------------------------
// (the one-column OX_EPI_CG_M2 form sits at 81 registers: held to 80 = 6 waves per SIMD, where the OX_EPI_DOT form runs)
template <int NC, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NC == 1 && EPI == OX_EPI_CG_M2) ? 6 : 1, 8))) void k_spmv_ps(ox_sell A, const double *__restrict__ x,
                                                 double *__restrict__ y,
                                                 const double *__restrict__ dinv,
                                                 const double *__restrict__ aux,
                                                 double *__restrict__ partial,
                                                 const int *__restrict__ done_flag /* never null */,
                                                 const int32_t *__restrict__ slice_list, int n_list, OxEpiDinv ED) {
  constexpr int NV = (EPI == OX_EPI_NONE) ? 1 : ox_epi_nv(EPI, NC);
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // 16-B load from an 8-B aligned address
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __shared__ double red[4 * NV];
  __shared__ double dict[4 * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_sl = slice_list ? n_list : A.n_slices;
  const int ngroups_l = (n_sl + 3) >> 2;
  const int per = gridDim.x >> 3;
  const int chunk = (ngroups_l + 7) >> 3;
  const int g_begin = (blockIdx.x & 7) * chunk;
  const int g_end = min(ngroups_l, g_begin + chunk);
  int g = g_begin + (blockIdx.x >> 3);
  // ---- round A
  const int dn = *done_flag;
  int li = g * 4 + wave;
  bool valid = g < g_end && li < n_sl;
  int slice = valid ? (slice_list ? slice_list[li] : li) : 0;
  int64_t base = A.ps_ptr[slice], next = A.ps_ptr[slice + 1];  // bit 0: slice kept in the entry stream
  const int nd = A.n_dict;
  double *md = dict + wave * 256;
  {
    const double d0 = A.vdict[min(lane, nd - 1)];
    if (nd > 64) {
      const double d1 = A.vdict[min(lane + 64, nd - 1)], d2 = A.vdict[min(lane + 128, nd - 1)],
                   d3 = A.vdict[min(lane + 192, nd - 1)];
      md[lane + 64] = d1;
      md[lane + 128] = d2;
      md[lane + 192] = d3;
    }
    md[lane] = d0;
  }
  double s[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) s[i] = 0.0;
  if (dn) valid = false;
  while (valid) {
    const bool wide = (base & 1) != 0;
    const int last = (int)((base >> 1) & 7);  // slots of the last group that any row uses (0: a stream without the hint)
    base &= ~(int64_t)255;
    const int ng = (int)(((next & ~(int64_t)255) - base) >> 8);  // groups of 4 slots x 64 lanes
    const int64_t row = (int64_t)slice * 64 + lane;
    const int64_t rowc = min(row, A.n_rows - 1);
    double xe[NC], ae[NC], de = 1.0;  // epilogue operands: requested now, used after the products
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // (OX_EPI_DOT reads its x entry behind the products instead: the diagonal's gather left it in L1, and two
      // registers fewer through the rounds are 72 instead of 78 = 7 waves per SIMD: 65.6 -> 64.8 us per pressure iteration)
      xe[c] = (EPI == OX_EPI_BCGS_T || EPI == OX_EPI_BCGS_T5 || EPI == OX_EPI_CG_M2) ? x[rowc * NC + c] : 0.0;
      ae[c] = (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T5) ? aux[rowc * NC + c] : 0.0;
    }
    if (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T || EPI == OX_EPI_BCGS_T5) de = dinv[rowc];
    // (through its dictionary: 1 B per row, hot in the L2 -- the vector kernels read the same codes; requested here with
    // the other epilogue operands: fetched after the products instead it added two dependent rounds to every wave,
    // 37.2 -> 39.5 us)
    if (EPI == OX_EPI_CG_M2) de = ED.code ? ED.dict[ED.code[rowc]] : dinv[rowc];
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const u4 *__restrict__ cp = reinterpret_cast<const u4 *>(A.ps_code + base) + lane;
    const int2 *__restrict__ bp = reinterpret_cast<const int2 *>(A.ps_base) + (base >> 8);
    // one group: 4 slots = up to 8 entries of the row, 4 gathers of 16 B (NC = 1)
    // one group: 4 slots = up to 8 entries of the row, 4 gathers of 16 B per component; the fused
    // multiply-adds in the stored order of the entries
    // NS: slots of the group in use (the rest is padding in every row of the slice: zero coefficients, no gather
    // needed); a compile-time count keeps the group's gathers one straight-line batch
    auto group = [&](auto ns_tag, const u4 code, const int2 b) {
      constexpr int NS = decltype(ns_tag)::value;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const unsigned cj = code[j];
        const int col = ((cj & 0x8000u) ? b.y : b.x) + (int)(cj & 0x7fffu);
        const double va = md[(cj >> 16) & 0xffu], vb = md[cj >> 24];
        const double *xp = x + (size_t)col * NC;
        d2u xv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) xv[c] = *reinterpret_cast<const d2u *>(xp + 2 * c);
        // x[col][0..NC-1] then x[col+1][0..NC-1] are the 2*NC doubles of xv[0..NC-1]
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fma(va, (c & 1) ? xv[c >> 1].y : xv[c >> 1].x, acc[c]);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fma(vb, ((NC + c) & 1) ? xv[(NC + c) >> 1].y : xv[(NC + c) >> 1].x, acc[c]);
      }
    };
    if (ng > 0) {
      if (!wide) {
        // ---- rounds B, C.  The code loads come from the Infinity Cache / HBM (the stream does not fit the L2s):
        // ~2000 cycles a round, the longest link of the wave's chain.  The codes of the first THREE groups are
        // therefore requested together (a P1 slice has 2 or 3): a 3-group slice saves a whole round.  Gathers
        // stay two groups per turn (eight in flight per wave at 100 registers measured 25 % slower).
        auto tail = [&](const u4 cc, const int2 bq) {  // the slice's last group
          if (last == 1) group(std::integral_constant<int, 1>{}, cc, bq);
          else if (last == 2) group(std::integral_constant<int, 2>{}, cc, bq);
          else if (last == 3) group(std::integral_constant<int, 3>{}, cc, bq);
          else group(std::integral_constant<int, 4>{}, cc, bq);
        };
        int q0 = 0;
        if (ng <= 3) {
          const u4 c0 = __builtin_nontemporal_load(cp);
          const u4 c1 = __builtin_nontemporal_load(cp + (size_t)min(1, ng - 1) * 64);
          const u4 c2 = __builtin_nontemporal_load(cp + (size_t)(ng - 1) * 64);
          const int2 b0 = bp[0], b1 = bp[min(1, ng - 1)], b2 = bp[ng - 1];
          if (ng == 1) {
            tail(c0, b0);
          } else {
            group(std::integral_constant<int, 4>{}, c0, b0);
            if (ng == 2) {
              tail(c1, b1);
            } else {
              group(std::integral_constant<int, 4>{}, c1, b1);
              tail(c2, b2);
            }
          }
          q0 = ng;
        }
        for (int q = q0; q < ng; q += 2) {
          const int q1 = min(q + 1, ng - 1);
          const u4 ca = __builtin_nontemporal_load(cp + (size_t)q * 64);
          const u4 cb2 = __builtin_nontemporal_load(cp + (size_t)q1 * 64);
          const int2 ba = bp[q], bb = bp[q1];
          if (q + 1 < ng) {
            group(std::integral_constant<int, 4>{}, ca, ba);
            if (q + 2 < ng) group(std::integral_constant<int, 4>{}, cb2, bb);
            else tail(cb2, bb);
          } else {
            tail(ca, ba);
          }
        }
      } else {  // this slice's pair columns did not fit two 15-bit windows (rare): entry stream
        const int64_t eb = A.slice_ptr[slice];
        const int npair = (int)((A.slice_ptr[slice + 1] - eb) >> 7);
        const int2 *__restrict__ ecp = reinterpret_cast<const int2 *>(A.cols + eb) + lane;
        const unsigned short *__restrict__ vcp = reinterpret_cast<const unsigned short *>(A.vcode + eb) + lane;
        for (int k = 0; k < npair; ++k) {
          const unsigned c2 = vcp[(size_t)k * 64];
          const int2 c = ecp[(size_t)k * 64];
          const double va = md[c2 & 0xff], vb = md[c2 >> 8];
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(va, x[(size_t)c.x * NC + cc], acc[cc]);
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(vb, x[(size_t)c.y * NC + cc], acc[cc]);
        }
      }
    }
    if (row < A.n_rows) {
      if (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T || EPI == OX_EPI_BCGS_T5) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] *= de;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) y[row * NC + c] = acc[c];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (EPI == OX_EPI_DOT) s[c] = fma(x[row * NC + c], acc[c], s[c]);
        if (EPI == OX_EPI_BCGS_V) s[c] = fma(ae[c], acc[c], s[c]);
        if (EPI == OX_EPI_BCGS_T) {
          s[c] = fma(acc[c], acc[c], s[c]);
          s[NC + c] = fma(acc[c], xe[c], s[NC + c]);
        }
        if (EPI == OX_EPI_BCGS_T5) {
          s[c] = fma(acc[c], acc[c], s[c]);
          s[NC + c] = fma(acc[c], xe[c], s[NC + c]);
          s[2 * NC + c] = fma(ae[c], xe[c], s[2 * NC + c]);
          s[3 * NC + c] = fma(ae[c], acc[c], s[3 * NC + c]);
          s[4 * NC + c] = fma(xe[c], xe[c], s[4 * NC + c]);
        }
        if (EPI == OX_EPI_CG_M2) {
          s[c] = fma(xe[c], acc[c], s[c]);
          s[NC + c] = fma(acc[c], de * acc[c], s[NC + c]);
        }
      }
    }
    g += per;
    li = g * 4 + wave;
    valid = g < g_end && li < n_sl;
    if (valid) {
      slice = slice_list ? slice_list[li] : li;
      base = A.ps_ptr[slice];
      next = A.ps_ptr[slice + 1];
    }
  }
  if (EPI != OX_EPI_NONE) {
    ox_block_sum_256<NV>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) partial[(size_t)blockIdx.x * NV + i] = s[i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_spmv_win: y = A x with the x operands read from an LDS copy of the block's WINDOW (ox_sell.wb_* / wlist / wcode).
// The lane = row kernels above issue one gather wave-instruction per entry (two with three right-hand sides), and
// the texture path takes ~32 cycles for each whatever its width or footprint: they are bound by those gathers, not by
// bytes (profiles/r03_spmv_u128_counters.txt: TA 85-90 %, TD 87-98 % busy).  Here a block of 4 waves owns up to 8
// slices (512 rows) that lie close together in the mesh; the distinct columns of all their entries -- the window,
// 1.3-2.5 K dofs on a box mesh in brick order, against 14.6 K entries -- are gathered ONCE into LDS (coalesced list
// reads, |window| gathers), and every operand of the products is a ds_read through the entry's 16-bit window index.
// Same entries, same per-row order of fused multiply-adds as k_spmv: bit-identical results.
// The 8 slices are dealt to the 4 waves by a stored schedule (wb_waves) that balances their widths -- a P2 block
// holds one 66-wide vertex slice, four 28-wide and three 20-wide edge slices: 66 / 56 / 56 / 60 --, a block whose
// window exceeds w_cap (the launch's LDS budget) multiplies from the int32 columns like k_spmv.
// ---------------------------------------------------------------------------------------
template <int NC, int EPI, int VAR>
__global__ __launch_bounds__(256) void k_spmv_win(ox_sell A, const double *__restrict__ x, double *__restrict__ y,
                                                  const double *__restrict__ dinv, const double *__restrict__ aux,
                                                  double *__restrict__ partial, const int *__restrict__ done_flag,
                                                  int w_cap, OxEpiDinv ED, int wb0, int wbn) {
  constexpr int NV = (EPI == OX_EPI_NONE) ? 1 : ox_epi_nv(EPI, NC);
  extern __shared__ double xw[];  // [NC][w_cap]: one plane per right-hand side -- the 64 lanes of a wave mostly read
                                  // consecutive window entries: consecutive 8-byte words, no bank conflicts (the
                                  // interleaved [w_cap][3] form spent 48 % of its LDS cycles on conflicts)
  __shared__ double red[4 * NV];
  __shared__ double dict[(VAR & 4) ? 256 : 1];
  if (done_flag && *done_flag) return;
  if (VAR & 4) {
    if ((int)threadIdx.x < A.n_dict) dict[threadIdx.x] = A.vdict[threadIdx.x];
  }
  // blocks of one XCD: a contiguous eighth of the launch's window blocks [wb0, wb0 + wbn) -- all of them, or the
  // interior / the boundary ones of a mesh-partitioned operator (ox_sell.n_wb_interior)
  const int bl = ox_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wb0 + bl;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double s[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) s[i] = 0.0;
  const bool live = bl < wbn;
  int64_t w0 = 0;
  int W = 0;
  if (live) {
    w0 = A.wb_ptr[b];
    W = (int)(A.wb_ptr[b + 1] - w0);
  }
  const bool windowed = live && W <= w_cap;
  if (windowed) {  // fill: the list is read coalesced, x[list] gathered once per window entry
    const int32_t *__restrict__ wl = A.wlist + w0;
    for (int i0 = threadIdx.x; i0 < W; i0 += 1024) {
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = wl[min(i0 + u * 256, W - 1)];
      double v[4][NC];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v[u][cc] = x[(size_t)c[u] * NC + cc];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * 256 < W) {
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) xw[(size_t)cc * w_cap + (i0 + u * 256)] = v[u][cc];
        }
    }
  }
  __syncthreads();
  if (live) {
    const unsigned sched = A.wb_waves[b];
    const int32_t *__restrict__ sl = A.wb_slices + (size_t)b * 8;
    for (int j = 0; j < 8; ++j) {
      if ((int)((sched >> (2 * j)) & 3u) != wave) continue;
      const int slice = __builtin_amdgcn_readfirstlane(sl[j]);
      if (slice < 0) continue;
      const int64_t row = (int64_t)slice * 64 + lane;
      double acc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = 0.0;
      const int64_t base = A.slice_ptr[slice];
      const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);
      typedef double v2d __attribute__((ext_vector_type(2)));
      typedef int v2i __attribute__((ext_vector_type(2)));
      const double2 *__restrict__ vp = reinterpret_cast<const double2 *>(A.vals + base) + lane;
      const unsigned short *__restrict__ vcp =
          (VAR & 4) ? reinterpret_cast<const unsigned short *>(A.vcode + base) + lane : nullptr;
      auto load_vals = [&](int k) {
        double2 v;
        if (VAR & 4) {
          const unsigned cc = __builtin_nontemporal_load(vcp + (size_t)k * 64);
          v.x = dict[cc & 0xff];
          v.y = dict[cc >> 8];
        } else {
          const v2d vv = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(vp + (size_t)k * 64));
          v.x = vv.x;
          v.y = vv.y;
        }
        return v;
      };
      if (windowed) {
        // tile layout: one 8-byte load = the window indices of 2 pairs (4 entries) of this lane, one 4-byte load
        // their value codes; f64 values stay in the pair layout (16 bytes per pair).  Two tiles are in flight; the
        // products keep the stored order of the entries.
        typedef unsigned short v4h __attribute__((ext_vector_type(4)));
        typedef unsigned char v4b __attribute__((ext_vector_type(4)));
        const int64_t t0 = A.wt_ptr[slice];
        const int ntile = (npair + 1) >> 1;
        const v4h *__restrict__ hp = reinterpret_cast<const v4h *>(A.wcode) + t0 * 64 + lane;
        const v4b *__restrict__ bp = (VAR & 4) ? reinterpret_cast<const v4b *>(A.wvcode) + t0 * 64 + lane : nullptr;
        auto tile = [&](int t, auto full_tag) {
          constexpr bool FULL = decltype(full_tag)::value;
          const v4h d = __builtin_nontemporal_load(hp + (size_t)t * 64);
          double2 v[2];
          if (VAR & 4) {
            const v4b cc = __builtin_nontemporal_load(bp + (size_t)t * 64);
#pragma unroll
            for (int j = 0; j < 2; ++j) v[j].x = dict[cc[2 * j]], v[j].y = dict[cc[2 * j + 1]];
          } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int k = FULL ? 2 * t + j : min(2 * t + j, npair - 1);
              const v2d vv = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(vp + (size_t)k * 64));
              v[j].x = vv.x, v[j].y = vv.y;
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (FULL || 2 * t + j < npair) {
              const double *x0 = xw + d[2 * j];
              const double *x1 = xw + d[2 * j + 1];
#pragma unroll
              for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(v[j].x, x0[(size_t)cc * w_cap], acc[cc]);
#pragma unroll
              for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(v[j].y, x1[(size_t)cc * w_cap], acc[cc]);
            }
          }
        };
        const int nfull = npair >> 1;
#pragma unroll 4
        for (int t = 0; t < nfull; ++t) tile(t, std::true_type{});
        if (nfull < ntile) tile(nfull, std::false_type{});
      } else {
        const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
#pragma unroll 4
        for (int k = 0; k < npair; ++k) {
          const double2 v = load_vals(k);
          const v2i ci = __builtin_nontemporal_load(reinterpret_cast<const v2i *>(cp + (size_t)k * 64));
          const double *x0 = x + (size_t)ci.x * NC;
          const double *x1 = x + (size_t)ci.y * NC;
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(v.x, x0[cc], acc[cc]);
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) acc[cc] = fma(v.y, x1[cc], acc[cc]);
        }
      }
      if (row < A.n_rows) {
        if (EPI == OX_EPI_BCGS_V || EPI == OX_EPI_BCGS_T || EPI == OX_EPI_BCGS_T5) {
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
          if (EPI == OX_EPI_BCGS_T5) {
            const double xs = x[row * NC + c], ah = aux[row * NC + c];
            s[c] = fma(acc[c], acc[c], s[c]);
            s[NC + c] = fma(acc[c], xs, s[NC + c]);
            s[2 * NC + c] = fma(ah, xs, s[2 * NC + c]);
            s[3 * NC + c] = fma(ah, acc[c], s[3 * NC + c]);
            s[4 * NC + c] = fma(xs, xs, s[4 * NC + c]);
          }
          if (EPI == OX_EPI_CG_M2) {
            const double dd = ED.code ? ED.dict[ED.code[row]] : dinv[row];
            s[c] = fma(x[row * NC + c], acc[c], s[c]);
            s[NC + c] = fma(acc[c], dd * acc[c], s[NC + c]);
          }
        }
      }
    }
  }
  if (EPI != OX_EPI_NONE) {
    ox_block_sum_256<NV>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) partial[(size_t)blockIdx.x * NV + i] = s[i];
    }
  }
}

// a device int that is always 0: stands in for a null done flag so the kernel can load it unconditionally.
// One allocation per device for the whole process, zeroed on the caller's stream before its first use there.
#include <mutex>
static const int *ox_zero_flag(hipStream_t st) {
  static std::mutex mu;
  static int *z[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!z[dev]) {
    int *p = nullptr;
    if (hipMalloc(&p, 64) != hipSuccess) return nullptr;
    // ordered before the launch that follows on `st`; a later use from another stream happens after this
    // call has returned, and the flag is never written again
    if (hipMemsetAsync(p, 0, 64, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return nullptr;
    z[dev] = p;
  }
  return z[dev];
}

// bit 0: nontemporal matrix stream; bit 1: 16-bit column stream, bit 2: 1-byte value codes, bit 3: the
// pair-slot stream, bit 4: the LDS-window stream, each where the matrix carries them (measured: tools/spmv_bench.py).
// Every level multiplies the same entries in the same order: the switch never changes a result.
#define OX_SPMV_DEFAULT_VARIANT 31
// (a per-matrix field, ox_sell.levels, set through the ABI by whoever owns the matrix: the process-wide switch and its
// OX_SPMV_VARIANT environment variable of rounds 2-4 let one object's tuning call change another's schedule)
static inline int spmv_levels(const ox_sell *A) { return (A->levels & 32) ? (A->levels & 31) : OX_SPMV_DEFAULT_VARIANT; }

// LDS-window stream: window entries the kernel's LDS budget holds with ncomp right-hand sides (48 / 48 / 51 KB: three
// resident blocks per CU with three columns); ox_sell.w_cap overrides (tuning).  0: the matrix has no window stream, or
// the launch cannot use it (slice lists of a partitioned operator).
static int spmv_window_cap(const ox_sell *A, int ncomp, const int32_t *list) {
  // slice lists: only the interior / boundary halves of a partitioned operator whose window blocks are split alike
  if (list && !(A->ib_slices && (list == A->ib_slices || list == A->ib_slices + A->n_interior))) return 0;
  if (!(spmv_levels(A) & 16) || !A->wcode || !A->wt_ptr || !A->wlist || !A->wb_ptr || !A->wb_slices ||
      !A->wb_waves || A->n_wblocks <= 0 || ncomp < 1 || ncomp > 3)
    return 0;
  const int dflt = ncomp == 1 ? 6144 : (ncomp == 2 ? 3072 : 2176);
  const int cap = A->w_cap > 0 ? A->w_cap : dflt;
  return A->w_max < cap ? A->w_max : cap;
}
static inline int spmv_window_grid_n(int nblocks) { return (nblocks + 7) & ~7; }
// window blocks [first, first + count) of a launch over `list`: all, the interior or the boundary ones
static inline void spmv_window_range(const ox_sell *A, const int32_t *list, int n_list, int *first, int *count) {
  if (!list) *first = 0, *count = A->n_wblocks;
  else if (list == A->ib_slices && n_list == A->n_interior) *first = 0, *count = A->n_wb_interior;
  else *first = A->n_wb_interior, *count = A->n_wblocks - A->n_wb_interior;
}

// the Jacobi diagonal of the OX_EPI_CG_M2 epilogue through its value dictionary (ox_ksp.hip sets it around its mat-vecs;
// one call at a time per process: include/oasisx_hip.h)
static thread_local OxEpiDinv g_epi_dinv{nullptr, nullptr};  // (set and read by the same host thread, around its mat-vecs)
void ox_spmv_set_epilogue_dinv(const uint8_t *code, const double *dict) { g_epi_dinv = OxEpiDinv{code, dict}; }

static int spmv_launch_list(const ox_sell *A, const double *x, double *y, int ncomp, int epi,
                            const double *dinv, const double *aux, double *partial, const int *done,
                            hipStream_t st, const int32_t *list, int n_list) {
  const int nblk = list ? ox_spmv_blocks_n(n_list) : ox_spmv_blocks(A);
  if (nblk == 0) return 0;
  const int lv = spmv_levels(A);
  int var = (A->cols16 && A->cbase) ? (lv & 7) : (lv & 1);
  // bit 3 off: ignore the pair-slot stream (tools/spmv_bench.py A/B)
  const bool pairs = (lv & 8) && A->ps_ptr && A->ps_code && A->ps_base;
  // value codes ride on the 16-bit column stream (one kernel family: 7 = all three)
  if ((var & 6) != 6 || !A->vcode || !A->vdict || A->n_dict < 1 || A->n_dict > 256) var &= 3;
  else var = 7;
  const int w_cap = pairs ? 0 : spmv_window_cap(A, ncomp, list);
  if (w_cap > 0) {
    const bool codes = var == 7 && A->wvcode;  // a dictionary matrix without tiled value codes multiplies its f64 values
    int wb0 = 0, wbn = 0;
    spmv_window_range(A, list, n_list, &wb0, &wbn);
    if (wbn == 0) return 0;
    const int wgrid = spmv_window_grid_n(wbn);
    const size_t lds = (size_t)w_cap * ncomp * sizeof(double);
#define OX_WIN_CASE(NC, E)                                                                                          \
  if (ncomp == NC && epi == E) {                                                                                    \
    auto go = [&](auto kern) -> int {                                                                               \
      if (lds > 48 * 1024)                                                                                          \
        OX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds));                                                                      \
      if (ox_prof_on) ox_prof_start(OX_TAG_SPMV(NC, E), st, A->n_rows);                                             \
      hipLaunchKernelGGL(kern, dim3(wgrid), dim3(256), lds, st, *A, x, y, dinv, aux, partial, done, w_cap, g_epi_dinv, wb0, wbn); \
      if (ox_prof_on) ox_prof_stop(st);                                                                             \
      OX_LAUNCH_CHECK();                                                                                            \
      return 0;                                                                                                     \
    };                                                                                                              \
    return codes ? go(k_spmv_win<NC, E, 5>) : go(k_spmv_win<NC, E, 1>);                                             \
  }
#define OX_WIN_NC(NC)                                                                                   \
  OX_WIN_CASE(NC, OX_EPI_NONE) OX_WIN_CASE(NC, OX_EPI_DOT) OX_WIN_CASE(NC, OX_EPI_BCGS_V) OX_WIN_CASE(NC, OX_EPI_BCGS_T) \
  OX_WIN_CASE(NC, OX_EPI_BCGS_T5) OX_WIN_CASE(NC, OX_EPI_CG_M2)
    OX_WIN_NC(1) OX_WIN_NC(2) OX_WIN_NC(3)
#undef OX_WIN_NC
#undef OX_WIN_CASE
  }
#define OX_SPMV_LAUNCH(NC, E, V)                                                                   \
  hipLaunchKernelGGL((k_spmv<NC, E, V>), dim3(nblk), dim3(256), 0, st, *A, x, y, dinv, aux, partial, done, list, n_list, g_epi_dinv)
#define OX_SPMV_CASE(NC, E)                                                                     \
  if (ncomp == NC && epi == E) {                                                                \
    if (ox_prof_on) ox_prof_start(OX_TAG_SPMV(NC, E), st, A->n_rows);                                      \
    if (var == 7 && pairs) {                                                                    \
      const int *dz = done ? done : ox_zero_flag(st);                                             \
      if (!dz) OX_FAIL("ox_spmv: no device flag");                                              \
      hipLaunchKernelGGL((k_spmv_ps<NC, E>), dim3(nblk), dim3(256), 0, st, *A, x, y, dinv, aux, partial, dz, list, n_list, g_epi_dinv); \
    } else if (var == 7) OX_SPMV_LAUNCH(NC, E, 7);                                              \
    else if (var == 3) OX_SPMV_LAUNCH(NC, E, 3);                                                \
    else if (var == 2) OX_SPMV_LAUNCH(NC, E, 2);                                                \
    else if (var == 1) OX_SPMV_LAUNCH(NC, E, 1);                                                \
    else OX_SPMV_LAUNCH(NC, E, 0);                                                              \
    if (ox_prof_on) ox_prof_stop(st);                                                           \
    OX_LAUNCH_CHECK();                                                                          \
    return 0;                                                                                   \
  }
#define OX_SPMV_NC(NC)               \
  OX_SPMV_CASE(NC, OX_EPI_NONE)      \
  OX_SPMV_CASE(NC, OX_EPI_DOT)       \
  OX_SPMV_CASE(NC, OX_EPI_BCGS_V)    \
  OX_SPMV_CASE(NC, OX_EPI_BCGS_T)    \
  OX_SPMV_CASE(NC, OX_EPI_BCGS_T5)   \
  OX_SPMV_CASE(NC, OX_EPI_CG_M2)
  OX_SPMV_NC(1) OX_SPMV_NC(2) OX_SPMV_NC(3)
#undef OX_SPMV_NC
#undef OX_SPMV_CASE
  OX_FAIL("ox_spmv: unsupported ncomp=%d epi=%d", ncomp, epi);
}

int ox_spmv_launch(const ox_sell *A, const double *x, double *y, int ncomp, int epi,
                   const double *dinv, const double *aux, double *partial, const int *done,
                   hipStream_t st) {
  return spmv_launch_list(A, x, y, ncomp, epi, dinv, aux, partial, done, st, nullptr, 0);
}

// Overlapped mat-vec of a partitioned operator: start the halo exchange, multiply the interior slices while it is in
// flight, the boundary slices after it has landed -- or exchange, then multiply.  A field of the PLAN (ox_dist.overlap,
// ox_dist_set_overlap; the host layer reads OX_HALO_OVERLAP once and sets it): -1 = the transport's default: overlap on
// the xGMI-window transport and on the callback transports (both exercised by the multi-rank rehearsals on one GPU);
// exchange-then-multiply on RCCL plans -- there the overlap puts pack + ncclSend/ncclRecv on a side stream between two
// events while the main stream may issue an ncclAllReduce on the same communicator, an ordering that has not run between
// two real GPUs yet (no multi-GPU node in rounds 1-5): the conservative form stays the default until it has.
static bool ox_overlap_on(const ox_sell *A, const ox_dist *dist) {
  if (!(dist && dist->n_peers > 0 && A->ib_slices && A->n_interior > 0)) return false;
  if (dist->overlap >= 0) return dist->overlap != 0;
  return !(dist->comm && !dist->p2p && !dist->halo_cb);  // RCCL plan: off by default
}

// partial-sum rows (= grid blocks) of the launch over `list` (nullptr: the whole operator)
static int spmv_parts_of(const ox_sell *A, int ncomp, const int32_t *list, int n_list) {
  const bool pairs = (spmv_levels(A) & 8) && A->ps_ptr && A->ps_code && A->ps_base;
  if (!pairs && spmv_window_cap(A, ncomp, list) > 0) {
    int wb0 = 0, wbn = 0;
    spmv_window_range(A, list, n_list, &wb0, &wbn);
    return wbn == 0 ? 0 : spmv_window_grid_n(wbn);
  }
  return list ? ox_spmv_blocks_n(n_list) : ox_spmv_blocks(A);
}

int ox_spmv_dist_nparts(const ox_sell *A, const ox_dist *dist, int ncomp) {
  if (!ox_overlap_on(A, dist)) return spmv_parts_of(A, ncomp, nullptr, 0);
  return spmv_parts_of(A, ncomp, A->ib_slices, A->n_interior) +
         spmv_parts_of(A, ncomp, A->ib_slices + A->n_interior, A->n_slices - A->n_interior);
}

// y = A x with the ghost block of x refreshed: halo exchange started, interior slices multiplied
// while it is in flight, boundary slices after it has landed (or exchange-then-multiply without the split)
int ox_spmv_dist(const ox_sell *A, double *x, double *y, int ncomp, int epi, const double *dinv, const double *aux,
                 double *partial, const int *done, const ox_dist *dist, hipStream_t st) {
  if (!ox_overlap_on(A, dist)) {
    if (dist && ox_halo_forward_impl(dist, x, ncomp, st)) return -1;
    return ox_spmv_launch(A, x, y, ncomp, epi, dinv, aux, partial, done, st);
  }
  if (ox_prof_on) ox_prof_start(OX_TAG_HALO, st, ncomp);
  if (ox_halo_begin_impl(dist, x, ncomp, st)) return -1;
  if (ox_prof_on) ox_prof_stop(st);
  const int nv = ox_epi_nv(epi, ncomp);
  const int nb_int = spmv_parts_of(A, ncomp, A->ib_slices, A->n_interior);
  if (spmv_launch_list(A, x, y, ncomp, epi, dinv, aux, partial, done, st, A->ib_slices, A->n_interior)) return -1;
  if (ox_halo_end_impl(dist, x, ncomp, st)) return -1;
  return spmv_launch_list(A, x, y, ncomp, epi, dinv, aux, partial ? partial + (size_t)nb_int * nv : nullptr, done, st,
                          A->ib_slices + A->n_interior, A->n_slices - A->n_interior);
}

extern "C" int ox_spmv(const ox_sell *A, const double *x, double *y, int ncomp,
                       const ox_dist *dist, void *stream) {
  if (!A || !x || !y) OX_FAIL("ox_spmv: null argument");
  if (ncomp < 1 || ncomp > OX_MAXC) OX_FAIL("ox_spmv: ncomp=%d out of range", ncomp);
  hipStream_t st = ox_stream(stream);
  return ox_spmv_dist(A, const_cast<double *>(x), y, ncomp, OX_EPI_NONE, nullptr, nullptr, nullptr, nullptr, dist, st);
}

// ---------------------------------------------------------------------------------------
// 16-bit column stream.  In SELL-64 row order the k-th entries of a slice's 64 rows lie close
// together (same stencil offset of neighbouring rows), or in two such groups where the slice mixes
// rows next to a far block of the numbering with rows that are not.  Per pair of storage columns
// (128 entries) two int32 bases + 16-bit codes (bit 15 = which base, low 15 bits = offset)
// reproduce the int32 columns exactly.  A slice with a pair that needs a third base keeps its
// int32 columns (cbase[first pair][0] = -1).  One wave per slice; set-up time only.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int ox_wave_min_all(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int ox_wave_max_all(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void k_compress_cols(ox_sell A, uint16_t *__restrict__ cols16,
                                                       int32_t *__restrict__ cbase,
                                                       unsigned long long *n_ok) {
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= A.n_slices) return;
  const int64_t base = A.slice_ptr[slice];
  const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);
  const int2 *cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
  ushort2 *hp = reinterpret_cast<ushort2 *>(cols16 + base) + lane;
  int2 *bp = reinterpret_cast<int2 *>(cbase) + (base >> 7);
  const int BIG = 0x7fffffff;
  bool ok = true;
  for (int k = 0; k < npair; ++k) {
    const int2 c = cp[(size_t)k * 64];
    const int lo = ox_wave_min_all(min(c.x, c.y));
    // second group: everything the first base cannot reach
    const bool fx = c.x - lo >= 32768, fy = c.y - lo >= 32768;
    const int lo2 = ox_wave_min_all(min(fx ? c.x : BIG, fy ? c.y : BIG));
    const int hi2 = ox_wave_max_all(max(fx ? c.x : -1, fy ? c.y : -1));
    const bool fits = (lo2 == BIG) || (hi2 - lo2 < 32768);
    ok = ok && fits;
    ushort2 d;
    d.x = !fits ? 0 : (unsigned short)(fx ? (0x8000 | (c.x - lo2)) : (c.x - lo));
    d.y = !fits ? 0 : (unsigned short)(fy ? (0x8000 | (c.y - lo2)) : (c.y - lo));
    hp[(size_t)k * 64] = d;
    if (lane == 0) bp[k] = make_int2(lo, lo2 == BIG ? lo : lo2);
  }
  if (lane == 0 && npair > 0) {
    if (!ok) bp[0].x = -1;
    else if (n_ok) atomicAdd(n_ok, (unsigned long long)npair * 128ull);
  }
}

extern "C" int ox_sell_compress_cols(const ox_sell *A, uint16_t *cols16, int32_t *cbase,
                                     int64_t *n_compressed, void *stream) {
  if (!A || !cols16 || !cbase) OX_FAIL("ox_sell_compress_cols: null argument");
  hipStream_t st = ox_stream(stream);
  if (n_compressed) *n_compressed = 0;
  if (A->n_slices == 0) return 0;
  unsigned long long *cnt = nullptr;
  if (n_compressed) {
    OX_HIP(hipMalloc(&cnt, sizeof(*cnt)));
    OX_HIP(hipMemsetAsync(cnt, 0, sizeof(*cnt), st));
  }
  hipLaunchKernelGGL(k_compress_cols, dim3((A->n_slices + 3) / 4), dim3(256), 0, st, *A, cols16, cbase, cnt);
  OX_LAUNCH_CHECK();
  if (n_compressed) {
    unsigned long long h = 0;
    OX_HIP(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    OX_HIP(hipStreamSynchronize(st));
    OX_HIP(hipFree(cnt));
    *n_compressed = (int64_t)h;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// Pre-assembled rectangular operators (reference low_memory_version = False,
// fracstep.py:499-502, 540-542, 642): one SELL pattern, GD values per entry ([slot][GD]).
//   S2V: y[row][d] = base[row][d] + scale * sum_k vals[k][d] * x[col_k]        (P_i ps, G_i dp)
//   V2S: y[row]    = base[row] + scale * sum_k sum_d vals[k][d] * x[col_k][d]  (sum_i D_i u_i)
// One pass reads the columns once for all GD components (4 + 8 GD bytes per entry).
// (Nontemporal 16-B loads + unroll 4, which help k_spmv, measured 30 % SLOWER here.)
// ---------------------------------------------------------------------------------------
template <int GD, int V2S, bool DICT>
__global__ __launch_bounds__(256) void k_spmv_multi(ox_sell A, const double *__restrict__ x,
                                                    const double *__restrict__ base, double scale,
                                                    double *__restrict__ y) {
  // DICT: the GD values of an entry as one packed uint32 of 1-byte codes into a dictionary of
  // <= 256 doubles (la.MultiSellMatrix.freeze) + the 16-bit column stream: 6 B per entry
  // instead of 4 + 8 GD.
  __shared__ double dict[DICT ? 256 : 1];
  if constexpr (DICT) {
    if ((int)threadIdx.x < A.n_dict) dict[threadIdx.x] = A.vdict[threadIdx.x];
    __syncthreads();
  }
  const int ngroups = (A.n_slices + 3) >> 2;
  const int chunk = (ngroups + 7) >> 3;
  const int g = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);  // XCD-chunked, as k_spmv
  if ((int)(blockIdx.x >> 3) >= chunk || g >= ngroups) return;
  const int slice = __builtin_amdgcn_readfirstlane(g * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (slice >= A.n_slices) return;
  const int64_t row = (int64_t)slice * 64 + lane;
  const int64_t sbase = A.slice_ptr[slice];
  const int npair = (int)((A.slice_ptr[slice + 1] - sbase) >> 7);
  const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + sbase) + lane;
  double acc[GD];
#pragma unroll
  for (int d = 0; d < GD; ++d) acc[d] = 0.0;
  auto mac = [&](const double (&v)[2 * GD], int c0, int c1) {
    if (V2S) {
      const double *x0 = x + (size_t)c0 * GD, *x1 = x + (size_t)c1 * GD;
#pragma unroll
      for (int d = 0; d < GD; ++d) acc[0] = fma(v[GD + d], x1[d], fma(v[d], x0[d], acc[0]));
    } else {
      const double x0 = x[c0], x1 = x[c1];
#pragma unroll
      for (int d = 0; d < GD; ++d) acc[d] = fma(v[GD + d], x1, fma(v[d], x0, acc[d]));
    }
  };
  if constexpr (DICT) {
    typedef unsigned short v2h __attribute__((ext_vector_type(2)));
    const uint2 *__restrict__ cc = reinterpret_cast<const uint2 *>(A.vcode) + (sbase >> 1) + lane;
    const int2 *__restrict__ cb = reinterpret_cast<const int2 *>(A.cbase) + (sbase >> 7);
    const v2h *__restrict__ hp = reinterpret_cast<const v2h *>(A.cols16 + sbase) + lane;
    const bool c16 = cb[0].x >= 0;  // wave-uniform: this slice has 16-bit column codes
#pragma unroll 4
    for (int k = 0; k < npair; ++k) {
      const uint2 code = cc[(size_t)k * 64];
      int c0, c1;
      if (c16) {
        const v2h d = hp[(size_t)k * 64];
        const int2 b = cb[k];
        const int dx = d.x, dy = d.y;
        c0 = ((dx & 0x8000) ? b.y : b.x) + (dx & 0x7fff);
        c1 = ((dy & 0x8000) ? b.y : b.x) + (dy & 0x7fff);
      } else {
        const int2 c = cp[(size_t)k * 64];
        c0 = c.x, c1 = c.y;
      }
      double v[2 * GD];
#pragma unroll
      for (int d = 0; d < GD; ++d) {
        v[d] = dict[(code.x >> (8 * d)) & 0xff];
        v[GD + d] = dict[(code.y >> (8 * d)) & 0xff];
      }
      mac(v, c0, c1);
    }
  } else {
    const double *__restrict__ vp = A.vals + (size_t)sbase * GD + (size_t)lane * 2 * GD;
#pragma unroll 2
    for (int k = 0; k < npair; ++k) {
      const int2 c = cp[(size_t)k * 64];
      const double *vv = vp + (size_t)k * 128 * GD;  // 2 slots x GD values, contiguous per lane
      double v[2 * GD];
#pragma unroll
      for (int d = 0; d < 2 * GD; ++d) v[d] = vv[d];
      mac(v, c.x, c.y);
    }
  }
  if (row < A.n_rows) {
    if (V2S) {
      y[row] = scale * acc[0] + (base ? base[row] : 0.0);
    } else {
#pragma unroll
      for (int d = 0; d < GD; ++d) y[row * GD + d] = fma(scale, acc[d], base ? base[row * GD + d] : 0.0);
    }
  }
}

extern "C" int ox_spmv_multi(int v2s, int gdim, const ox_sell *A, const double *x, const double *base,
                             double scale, double *y, const ox_dist *dist, void *stream) {
  if (!A || !x || !y) OX_FAIL("ox_spmv_multi: null argument");
  if (gdim != 2 && gdim != 3) OX_FAIL("ox_spmv_multi: gdim=%d", gdim);
  hipStream_t st = ox_stream(stream);
  if (dist && ox_halo_forward_impl(dist, const_cast<double *>(x), v2s ? gdim : 1, st)) return -1;
  const int nblk = ox_spmv_blocks(A);
  if (nblk == 0) return 0;
  const int tag = v2s ? OX_TAG_RECT_V2S : OX_TAG_RECT_S2V;
  const bool dict = A->vcode && A->vdict && A->cols16 && A->cbase && A->n_dict >= 1 && A->n_dict <= 256 &&
                    (spmv_levels(A) & 4);
  if (ox_prof_on) ox_prof_start(tag, st, A->n_rows);
#define OX_MULTI(GD, V, D) \
  hipLaunchKernelGGL((k_spmv_multi<GD, V, D>), dim3(nblk), dim3(256), 0, st, *A, x, base, scale, y)
  if (dict) {
    if (gdim == 2 && v2s) OX_MULTI(2, 1, true);
    if (gdim == 2 && !v2s) OX_MULTI(2, 0, true);
    if (gdim == 3 && v2s) OX_MULTI(3, 1, true);
    if (gdim == 3 && !v2s) OX_MULTI(3, 0, true);
  } else {
    if (gdim == 2 && v2s) OX_MULTI(2, 1, false);
    if (gdim == 2 && !v2s) OX_MULTI(2, 0, false);
    if (gdim == 3 && v2s) OX_MULTI(3, 1, false);
    if (gdim == 3 && !v2s) OX_MULTI(3, 0, false);
  }
#undef OX_MULTI
  if (ox_prof_on) ox_prof_stop(st);
  OX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Deterministic final reduction of per-block partials: sums[i] = sum_p partial[p*nv+i],
// always in the same order (no float atomics anywhere on the path).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(OX_RED_THREADS) void k_reduce_partials(
    const double *__restrict__ partial, int nparts, int nv, double *__restrict__ sums) {
  __shared__ double red[16 * OX_MAX_NV];
  double v[OX_MAX_NV];
  ox_gather_partials(partial, nparts, nv, v);
  ox_block_sum_wide(v, nv, red);
  if (threadIdx.x == 0)
    for (int i = 0; i < nv; ++i) sums[i] = v[i];
}

int ox_reduce_partials(const double *partial, int nparts, int nv, double *sums, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(ox_red_threads(nparts)), 0, st, partial, nparts, nv, sums);
  OX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// BLAS-1
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_axpby(int64_t n, double a, const double *x, double b,
                                               const double *y, double *z) {
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
    double2 r;
    if (b != 0.0) {
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      const double2 yv = reinterpret_cast<const double2 *>(y)[i];
      r.x = a * xv.x + b * yv.x;
      r.y = a * xv.y + b * yv.y;
    } else {
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      r.x = a * xv.x;
      r.y = a * xv.y;
    }
    reinterpret_cast<double2 *>(z)[i] = r;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    z[i] = (b != 0.0) ? a * x[i] + b * y[i] : a * x[i];
  }
}

extern "C" int ox_axpby(int64_t n, double a, const double *x, double b, const double *y, double *z,
                        void *stream) {
  if (n <= 0) return 0;
  if (!x || !z || (b != 0.0 && !y)) OX_FAIL("ox_axpby: null argument");
  const int nblk = ox_vec_blocks(n);
  hipLaunchKernelGGL(k_axpby, dim3(nblk), dim3(256), 0, ox_stream(stream), n, a, x, b, y, z);
  OX_LAUNCH_CHECK();
  return 0;
}

template <int NC>
__global__ __launch_bounds__(256) void k_dot(int64_t n_rows, const double *__restrict__ x,
                                             const double *__restrict__ y,
                                             double *__restrict__ partial) {
  __shared__ double red[4 * NC];
  double s[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) s[c] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += stride) {
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = fma(x[r * NC + c], y[r * NC + c], s[c]);
  }
  ox_block_sum_256<NC>(s, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) partial[(size_t)blockIdx.x * NC + c] = s[c];
  }
}

static thread_local double *g_scratch = nullptr;       // device scratch for ox_dot / ox_remove_mean (per host thread)
static thread_local double *g_scratch_host = nullptr;  // pinned
static int ox_scratch_init() {
  if (!g_scratch) {
    OX_HIP(hipMalloc(&g_scratch, sizeof(double) * (OX_VEC_MAX_BLOCKS * 4 + 16)));
    OX_HIP(hipHostMalloc(&g_scratch_host, sizeof(double) * 16));
  }
  return 0;
}

extern "C" int ox_dot(int64_t n_rows, int ncomp, const double *x, const double *y, double *out_host,
                      const ox_dist *dist, void *stream) {
  if (ncomp < 1 || ncomp > OX_MAXC) OX_FAIL("ox_dot: ncomp=%d", ncomp);
  if (ox_scratch_init()) return -1;
  hipStream_t st = ox_stream(stream);
  const int nblk = ox_vec_blocks(n_rows > 0 ? n_rows : 1);
  double *partial = g_scratch, *sums = g_scratch + OX_VEC_MAX_BLOCKS * 4;
  switch (ncomp) {
    case 1: hipLaunchKernelGGL(k_dot<1>, dim3(nblk), dim3(256), 0, st, n_rows, x, y, partial); break;
    case 2: hipLaunchKernelGGL(k_dot<2>, dim3(nblk), dim3(256), 0, st, n_rows, x, y, partial); break;
    default: hipLaunchKernelGGL(k_dot<3>, dim3(nblk), dim3(256), 0, st, n_rows, x, y, partial); break;
  }
  OX_LAUNCH_CHECK();
  if (ox_reduce_partials(partial, nblk, ncomp, sums, st)) return -1;
  if (dist && ox_allreduce_impl(dist, sums, ncomp, st)) return -1;
  OX_HIP(hipMemcpyAsync(g_scratch_host, sums, sizeof(double) * ncomp, hipMemcpyDeviceToHost, st));
  OX_HIP(hipStreamSynchronize(st));
  for (int c = 0; c < ncomp; ++c) out_host[c] = g_scratch_host[c];
  return 0;
}

// x -= (sum w*x)/wsum   (w == NULL: plain sum)
__global__ __launch_bounds__(256) void k_wsum(int64_t n, const double *__restrict__ x,
                                              const double *__restrict__ w,
                                              double *__restrict__ partial) {
  __shared__ double red[4];
  double s[1] = {0.0};
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    s[0] += w ? w[i] * x[i] : x[i];
  ox_block_sum_256<1>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(256) void k_shift(int64_t n, double *x, const double *sum,
                                               double inv_wsum) {
  const double m = sum[0] * inv_wsum;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] -= m;
}

extern "C" int ox_remove_mean(int64_t n, int64_t n_apply, double *x, const double *w, double wsum,
                              const ox_dist *dist, void *stream) {
  if (n_apply < n) n_apply = n;
  if (n_apply <= 0) return 0;
  if (ox_scratch_init()) return -1;
  if (wsum == 0.0) OX_FAIL("ox_remove_mean: wsum == 0");
  hipStream_t st = ox_stream(stream);
  const int nblk = ox_vec_blocks(n > 0 ? n : 1);
  double *partial = g_scratch, *sums = g_scratch + OX_VEC_MAX_BLOCKS * 4;
  hipLaunchKernelGGL(k_wsum, dim3(nblk), dim3(256), 0, st, n, x, w, partial);
  OX_LAUNCH_CHECK();
  if (ox_reduce_partials(partial, nblk, 1, sums, st)) return -1;
  if (dist && ox_allreduce_impl(dist, sums, 1, st)) return -1;
  hipLaunchKernelGGL(k_shift, dim3(ox_vec_blocks(n_apply)), dim3(256), 0, st, n_apply, x, sums, 1.0 / wsum);
  OX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Boundary conditions
// ---------------------------------------------------------------------------------------
__global__ void k_set_bc(double *b, const int32_t *dofs, const double *g, int64_t n, int ncomp,
                         int comp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[(int64_t)dofs[i] * ncomp + comp] = g[i];
}

extern "C" int ox_set_bc(double *b, const int32_t *dofs, const double *g, int64_t n, int ncomp,
                         int comp, void *stream) {
  if (n <= 0) return 0;
  if (!b || !dofs || !g) OX_FAIL("ox_set_bc: null argument");
  if (comp < 0 || comp >= ncomp) OX_FAIL("ox_set_bc: comp=%d ncomp=%d", comp, ncomp);
  hipLaunchKernelGGL(k_set_bc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ox_stream(stream),
                     b, dofs, g, n, ncomp, comp);
  OX_LAUNCH_CHECK();
  return 0;
}

__global__ void k_scatter_add(double *b, const int32_t *rows, const double *y, int64_t n, int ncomp,
                              int comp, double scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[(int64_t)rows[i] * ncomp + comp] += scale * y[i];
}

extern "C" int ox_scatter_add(double *b, const int32_t *rows, const double *y, int64_t n, int ncomp,
                              int comp, double scale, void *stream) {
  if (n <= 0) return 0;
  if (!b || !rows || !y) OX_FAIL("ox_scatter_add: null argument");
  if (comp < 0 || comp >= ncomp) OX_FAIL("ox_scatter_add: comp=%d ncomp=%d", comp, ncomp);
  hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ox_stream(stream),
                     b, rows, y, n, ncomp, comp, scale);
  OX_LAUNCH_CHECK();
  return 0;
}

__device__ __forceinline__ int64_t ox_entry(const int64_t base, int k, int lane) {
  return base + (int64_t)(k / OX_KV) * (64 * OX_KV) + lane * OX_KV + (k % OX_KV);
}

// One wave per 64 listed rows, each lane its row: the (k, k+1) pairs of a row are 16 B apart-aligned, so a lane
// stores 16 B at a time.  au (optional): the identity row's product with u1, (A u1)[row] = u1[row], for the
// mat-vec the fused assembly hands to the tentative-velocity solve (ox_assemble_first_au) -- one launch instead
// of a zero-rows launch plus two indexed copies per boundary condition.
__global__ __launch_bounds__(64) void k_zero_rows(ox_sell A, const int32_t *__restrict__ rows, int64_t n, double diag,
                                                  double *__restrict__ au, const double *__restrict__ u1, int ncomp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = rows[i];
  const int slice = row >> 6, lane = row & 63;
  const int64_t base = A.slice_ptr[slice];
  const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);
  const int2 *__restrict__ cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
  double2 *__restrict__ vp = reinterpret_cast<double2 *>(A.vals + base) + lane;
  bool placed = false;
  for (int k = 0; k < npair; ++k) {
    const int2 c = cp[(size_t)k * 64];
    double2 v;
    const bool d0 = (c.x == row) && !placed;  // first match = the real diagonal entry (padding repeats the row later)
    placed = placed || d0;
    const bool d1 = (c.y == row) && !placed;
    placed = placed || d1;
    v.x = d0 ? diag : 0.0;
    v.y = d1 ? diag : 0.0;
    vp[(size_t)k * 64] = v;
  }
  if (au) {
    for (int c = 0; c < ncomp; ++c) au[(size_t)row * ncomp + c] = diag * u1[(size_t)row * ncomp + c];
  }
}

extern "C" int ox_zero_rows_au(const ox_sell *A, const int32_t *rows, int64_t n, double diag, double *au,
                               const double *u1, int ncomp, void *stream) {
  if (n <= 0) return 0;
  if (!A || !rows || (au && !u1)) OX_FAIL("ox_zero_rows: null argument");
  if (au && (ncomp < 1 || ncomp > OX_MAXC)) OX_FAIL("ox_zero_rows_au: ncomp=%d", ncomp);
  hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ox_stream(stream), *A, rows, n, diag, au, u1,
                     ncomp);
  OX_LAUNCH_CHECK();
  return 0;
}

extern "C" int ox_zero_rows(const ox_sell *A, const int32_t *rows, int64_t n, double diag,
                            void *stream) {
  return ox_zero_rows_au(A, rows, n, diag, nullptr, nullptr, 1, stream);
}

__global__ __launch_bounds__(256) void k_zero_rows_cols(ox_sell A, const uint8_t *is_bc,
                                                        double diag) {
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= A.n_slices) return;
  const int64_t row = (int64_t)slice * 64 + lane;
  if (row >= A.n_rows) return;
  const int64_t base = A.slice_ptr[slice];
  const int width = (int)((A.slice_ptr[slice + 1] - base) >> 6);
  const bool rbc = is_bc[row];
  bool placed = false;
  for (int k = 0; k < width; ++k) {
    const int64_t e = ox_entry(base, k, lane);
    const int c = A.cols[e];
    if (rbc || is_bc[c]) {
      const bool isd = rbc && c == row && !placed;
      A.vals[e] = isd ? diag : 0.0;
      placed = placed || isd;
    }
  }
}

extern "C" int ox_zero_rows_cols(const ox_sell *A, const uint8_t *is_bc, double diag,
                                 void *stream) {
  if (!A || !is_bc) OX_FAIL("ox_zero_rows_cols: null argument");
  if (A->n_slices == 0) return 0;
  hipLaunchKernelGGL(k_zero_rows_cols, dim3((A->n_slices + 3) / 4), dim3(256), 0,
                     ox_stream(stream), *A, is_bc, diag);
  OX_LAUNCH_CHECK();
  return 0;
}

// dinv[row] = 1 / A[row,row]; the diagonal is the first entry whose column equals the row
// (padding entries repeat the row index with value 0 and always come later).
__global__ __launch_bounds__(256) void k_jacobi(ox_sell A, double *dinv) {
  const int b = ox_xcd_remap(blockIdx.x, gridDim.x);
  const int slice = b * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= A.n_slices) return;
  const int64_t row = (int64_t)slice * 64 + lane;
  const int64_t base = A.slice_ptr[slice];
  const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);
  const int2 *cp = reinterpret_cast<const int2 *>(A.cols + base) + lane;
  int kd = -1;
  for (int k = 0; k < npair; ++k) {
    const int2 c = cp[(size_t)k * 64];
    if (kd < 0 && c.x == row) kd = 2 * k;
    if (kd < 0 && c.y == row) kd = 2 * k + 1;
    if (__all(kd >= 0 || row >= A.n_rows)) break;  // every row of the slice has found its diagonal
  }
  if (row < A.n_rows) {
    const double d = kd >= 0 ? A.vals[ox_entry(base, kd, lane)] : 0.0;
    dinv[row] = d != 0.0 ? 1.0 / d : 1.0;
  }
}

extern "C" int ox_jacobi_setup(const ox_sell *A, double *dinv, void *stream) {
  if (!A || !dinv) OX_FAIL("ox_jacobi_setup: null argument");
  const int nblk = (A->n_slices + 3) / 4;  // one slice group per block (not the persistent grid)
  if (nblk == 0) return 0;
  hipLaunchKernelGGL(k_jacobi, dim3(nblk), dim3(256), 0, ox_stream(stream), *A, dinv);
  OX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Per-kernel timing with HIP events (enabled by bench.py around its timed region).
// ---------------------------------------------------------------------------------------
#include <vector>
bool ox_prof_on = false;
namespace {
struct ProfRec { int tag; long long key; hipEvent_t a, b; };
std::vector<ProfRec> g_prof;
std::vector<float> g_prof_ms;
size_t g_prof_used = 0, g_prof_n = 0;
constexpr int OX_PROF_NTAG = 256;
long long g_prof_seen[OX_PROF_NTAG];
int g_prof_every = 1;
bool g_prof_open = false;
}  // namespace

void ox_prof_start(int tag, hipStream_t st, long long key) {
  g_prof_open = false;
  // an event pair costs ~2 x 6 us of stream bubbles: time only every g_prof_every-th launch of a tag
  if (tag >= 0 && tag < OX_PROF_NTAG && (g_prof_seen[tag]++ % g_prof_every) != 0) return;
  if (g_prof_used >= g_prof.size()) return;
  ProfRec &r = g_prof[g_prof_used];
  r.tag = tag;
  r.key = key;
  if (hipEventRecord(r.a, st) == hipSuccess) g_prof_open = true;
}
void ox_prof_stop(hipStream_t st) {
  if (!g_prof_open) return;
  (void)hipEventRecord(g_prof[g_prof_used].b, st);
  ++g_prof_used;
  g_prof_open = false;
}

extern "C" int ox_profile_begin(int max_records, int sample_every) {
  if (max_records < 1) max_records = 1;
  g_prof_every = sample_every < 1 ? 1 : sample_every;
  for (int i = 0; i < OX_PROF_NTAG; ++i) g_prof_seen[i] = 0;
  while ((int)g_prof.size() < max_records) {
    ProfRec r{};
    OX_HIP(hipEventCreate(&r.a));
    OX_HIP(hipEventCreate(&r.b));
    g_prof.push_back(r);
  }
  g_prof_used = 0;
  g_prof_n = 0;
  ox_prof_on = true;
  return 0;
}

extern "C" int ox_profile_end(void) {
  ox_prof_on = false;
  OX_HIP(hipDeviceSynchronize());
  g_prof_ms.assign(g_prof_used, 0.f);
  for (size_t i = 0; i < g_prof_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) != hipSuccess) ms = -1.f;
    g_prof_ms[i] = ms;
  }
  g_prof_n = g_prof_used;
  return 0;
}

extern "C" int ox_profile_get(int tag, long long key, long long *count, double *total_ms) {
  if (tag < 0 || tag >= OX_PROF_NTAG) OX_FAIL("ox_profile_get: tag=%d", tag);
  long long n = 0;
  double t = 0.0;
  for (size_t i = 0; i < g_prof_n; ++i)
    if (g_prof[i].tag == tag && (key < 0 || g_prof[i].key == key) && g_prof_ms[i] >= 0.f) {
      ++n;
      t += g_prof_ms[i];
    }
  if (count) *count = n;
  if (total_ms) *total_ms = t;
  return 0;
}

// ---- tile layout of the LDS-window stream (ox_sell.wt_ptr): re-tile a per-slot array from the pair layout --------------
template <typename T>
__global__ __launch_bounds__(256) void k_window_retile(ox_sell A, const int64_t *__restrict__ wt_ptr, const T *__restrict__ src,
                                                       T *__restrict__ dst) {
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= A.n_slices) return;
  const int64_t base = A.slice_ptr[slice];
  const int npair = (int)((A.slice_ptr[slice + 1] - base) >> 7);
  const int64_t t0 = wt_ptr[slice];
  const int ntile = (npair + 1) >> 1;
  for (int t = 0; t < ntile; ++t)
    for (int j = 0; j < 2; ++j) {
      const int k = 2 * t + j;
      T a = 0, b = 0;
      if (k < npair) {
        a = src[base + (int64_t)k * 128 + lane * 2];
        b = src[base + (int64_t)k * 128 + lane * 2 + 1];
      }
      dst[((t0 + t) * 64 + lane) * 4 + 2 * j] = a;
      dst[((t0 + t) * 64 + lane) * 4 + 2 * j + 1] = b;
    }
}

extern "C" int ox_window_retile(const ox_sell *A, const int64_t *wt_ptr, const void *src, int elem_bytes, void *dst, void *stream) {
  if (!A || !wt_ptr || !src || !dst) OX_FAIL("ox_window_retile: null argument");
  if (elem_bytes != 1 && elem_bytes != 2) OX_FAIL("ox_window_retile: elem_bytes=%d", elem_bytes);
  const int nblk = (A->n_slices + 3) / 4;
  if (nblk == 0) return 0;
  if (elem_bytes == 1)
    hipLaunchKernelGGL(k_window_retile<uint8_t>, dim3(nblk), dim3(256), 0, ox_stream(stream), *A, wt_ptr,
                       static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst));
  else
    hipLaunchKernelGGL(k_window_retile<uint16_t>, dim3(nblk), dim3(256), 0, ox_stream(stream), *A, wt_ptr,
                       static_cast<const uint16_t *>(src), static_cast<uint16_t *>(dst));
  OX_LAUNCH_CHECK();
  return 0;
}

// ---- roctx ranges (the reference leaves tracing to PETSc's -log_view stages) -------------------
// ox_range_push / ox_range_pop bracket a phase of the time step on the HOST thread; rocprofv3 --marker-trace
// (with --kernel-trace) then segments the trace by phase.  librocprofiler-sdk-roctx is looked up at the first
// call (no link-time dependency); without it, or without a profiler attached, the calls do nothing.
#include <dlfcn.h>
namespace {
typedef int (*roctx_push_t)(const char *);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
int g_roctx_state = 0;  // 0 = not looked up, 1 = available, -1 = absent
void roctx_lookup() {
  g_roctx_state = -1;
  const char *names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
  for (const char *n : names) {
    void *h = dlopen(n, RTLD_LAZY | RTLD_GLOBAL);
    if (!h) continue;
    g_roctx_push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
    g_roctx_pop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
    if (g_roctx_push && g_roctx_pop) {
      g_roctx_state = 1;
      return;
    }
  }
}
}  // namespace

extern "C" int ox_range_push(const char *name) {
  if (g_roctx_state == 0) roctx_lookup();
  if (g_roctx_state == 1 && name) g_roctx_push(name);
  return 0;
}

extern "C" int ox_range_pop(void) {
  if (g_roctx_state == 1) g_roctx_pop();
  return 0;
}
