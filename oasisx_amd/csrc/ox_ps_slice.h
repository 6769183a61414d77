// Products of one 64-row slice of the pair-slot stream (ox_sell.ps_*): the inner part of k_spmv_ps (ox_spmv.hip), shared
// with the persistent CG kernel (ox_ksp.hip) so that both multiply the same entries in the same order.
#pragma once
#include <type_traits>

#include "ox_kernels.h"

// (x carries no __restrict__ here: the persistent kernel also stores to the vector it gathers from, between two calls)
// acc[c] += sum over the slice's entries of row (slice, lane); `base` = ps_ptr[slice] with its low 8 bits cleared,
// ng = groups of 4 slots x 64 lanes, wide / last = the hints of ps_ptr's low bits, md = the value dictionary (LDS).
template <int NC>
__device__ __forceinline__ void ox_ps_products(const ox_sell &A, const double *x, const double *md, int slice,
                                               int lane, int64_t base, int ng, bool wide, int last, double (&acc)[NC]) {
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // 16-B load from an 8-B aligned address
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
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
}
