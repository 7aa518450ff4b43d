// Device helpers shared by the contraction kernels (gemm.hip, chebtile.hip): vector types, the exact 3-way bf16 split
// of an fp32 value, the XCD-aware block swizzle and the LDS-only block barrier.
#pragma once
#include "p2m_common.h"

namespace p2m {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// x = h + m + l exactly, each a bf16 value (8 + 8 + 8 significand bits by truncation); slices in the HIGH halves
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2) & 0xFFFF0000u;
}
__device__ __forceinline__ unsigned pack_hi(unsigned lo_elem, unsigned hi_elem) { return (lo_elem >> 16) | hi_elem; }

// Four consecutive-k values -> their three slices, packed (two dwords = four bf16 per slice).  Bit for bit what
// split3 + pack_hi produce, with fewer VALU instructions: v_perm_b32 takes the HIGH halves of two registers in one
// instruction, so the slices need no masking before they are packed - only the two remainders need the masked value
// (5.5 instructions per element instead of ~6; the contractions are issue-bound, every staging instruction delays an
// MFMA of the co-resident wave).
__device__ __forceinline__ unsigned hi_pair(float lo_elem, float hi_elem) {   // (bits(lo) >> 16) | (bits(hi) & 0xFFFF0000)
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float low_part(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
// (Measured and rejected: the two remainders of a pair with one v_pk_add_f32 -- 36 instead of 44 VALU per 8 values, but
// 120 instead of 129 TFLOP/s on 3x128 -> 128: the packed op costs the issue slots of more than the two it replaces.)
__device__ __forceinline__ void split3_pack4(float x0, float x1, float x2, float x3, u32x2& ph, u32x2& pm, u32x2& pl) {
  const float r0 = low_part(x0), r1 = low_part(x1), r2 = low_part(x2), r3 = low_part(x3);
  const float s0 = low_part(r0), s1 = low_part(r1), s2 = low_part(r2), s3 = low_part(r3);
  ph = u32x2{hi_pair(x0, x1), hi_pair(x2, x3)};
  pm = u32x2{hi_pair(r0, r1), hi_pair(r2, r3)};
  pl = u32x2{hi_pair(s0, s1), hi_pair(s2, s3)};
}

// observed dispatch: block b runs on XCD b % 8 -> give each XCD a contiguous range of logical ids (nb a multiple of 8)
__device__ __forceinline__ int xcd_contiguous(int bid, int nb) {
  return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid;
}

// block barrier that orders LDS traffic only: global loads / stores of the wave stay in flight across it
__device__ __forceinline__ void lds_block_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace p2m
