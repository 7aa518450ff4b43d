// Device helpers shared by the contraction kernels (gemm.hip, chebtile.hip): vector types, the two ways an fp32 operand
// is cut into matrix-core slices (three bf16 slices, exact; two scaled fp16 slices, 22 bits), the XCD-aware block
// swizzle and the LDS-only block barrier.
#pragma once
#include "p2m_common.h"

namespace p2m {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Slice arithmetic of a contraction kernel: NS = 3 -> bf16 slices (P2M_ARITH_BF16X3), NS = 2 -> fp16 slices (P2M_ARITH_F16X2)
template <int NS> struct SliceFrag;
template <> struct SliceFrag<3> { typedef bf16x8 type; };
template <> struct SliceFrag<2> { typedef f16x8 type; };
template <int NS>
__device__ __forceinline__ floatx16 slice_mfma(typename SliceFrag<NS>::type a, typename SliceFrag<NS>::type b, floatx16 c) {
  if constexpr (NS == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

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

// ---- two fp16 slices (P2M_ARITH_F16X2) ------------------------------------------------------------------------------
// fp16 has 11 significand bits but only 5 exponent bits, so an operand tensor is first multiplied by a power of two
// chosen from an upper bound U of its magnitudes (an "amax word": the bits of a non-negative float >= max |x|, kept in
// device memory by whoever produced the tensor, p2m_amax* / the amax_out arguments): 2^s with U 2^s in [2^14, 2^15).
// Then  x 2^s = h + l + e,  h = fp16(x 2^s),  l = fp16(x 2^s - h),  |e| <= 2^-22 |x 2^s| for |x 2^s| >= 2^-3 and
// <= 2^-25 (absolute, = 2^-40 U) below: 22 significand bits relative to anything within 18 binades of the tensor's
// maximum.  A product keeps h h' + h l' + l h' (three MFMAs instead of six); the dropped l l' is <= 2^-22 |x y|.  The
// accumulator holds C 2^(sa + sb); the epilogue undoes it with one exact v_ldexp.
// headroom: extra binades for operands DERIVED from the bounded tensor inside the same launch (the Chebyshev planes L x,
// L2 x are bounded by (max row sum of |L|, |L2|) U, the pair sums by twice that): p2m_graph plane_bits.
__device__ __forceinline__ int slice_scale_exp(unsigned amax_bits, int headroom) {
  const int e = (int)((amax_bits >> 23) & 0xFFu);     // U < 2^(e - 126)
  if (amax_bits == 0u) return 0;                      // an all-zero tensor
  int s = 141 - e - headroom;                         // U 2^headroom 2^s < 2^15
  return s > 120 ? 120 : (s < -120 ? -120 : s);       // 2^s stays a normal float (denormal / huge operands lose bits, not sense)
}
__device__ __forceinline__ float exp2_int(int s) { return __uint_as_float((unsigned)(s + 127) << 23); }   // |s| <= 126

__device__ __forceinline__ unsigned pack_f16_rne(float a, float b) {      // v_cvt_pk_f16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
// Four consecutive-k values -> their two slices, packed.  8 VALU instructions per 4 values (round 4; 12 before: 4 v_mul,
// 2 v_cvt_pk_f16_f32, 4 v_fma_mix_f32, 2 v_cvt_pk_f16_f32): the mixed-precision fma writes its f16 result straight into one
// half of the destination, so   h = fp16(x sc)          is v_fma_mixlo_f16 / v_fma_mixhi_f16 (x, sc, 0)   and
//                               l = fp16(x sc - h)      is v_fma_mixlo_f16 / v_fma_mixhi_f16 (x, sc, -h as an f16 source)
// - the same values bit for bit: x sc is exact (sc is a power of two), the fp32 fma x sc - h is exact (h carries the top 11
// bits of x sc), and each result is rounded to fp16 once, to nearest even.  The contractions are issue-bound (every
// staging instruction delays an MFMA of the co-resident wave): a third fewer split instructions in every producer.
__device__ __forceinline__ void split2_pack4(float x0, float x1, float x2, float x3, float sc, u32x2& ph, u32x2& pl) {
  unsigned h01, h23, l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h01) : "v"(x0), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h01) : "v"(x1), "v"(sc));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h23) : "v"(x2), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h23) : "v"(x3), "v"(sc));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l01) : "v"(x0), "v"(sc), "v"(h01));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(x1), "v"(sc), "v"(h01));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l23) : "v"(x2), "v"(sc), "v"(h23));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(x3), "v"(sc), "v"(h23));
  ph = u32x2{h01, h23};
  pl = u32x2{l01, l23};
}
// both cuts behind one name: sl[0] = high slice ... sl[NS - 1] = low slice
template <int NS>
__device__ __forceinline__ void split_pack4(float x0, float x1, float x2, float x3, float sc, u32x2 (&sl)[NS]) {
  if constexpr (NS == 3) split3_pack4(x0, x1, x2, x3, sl[0], sl[1], sl[2]);
  else split2_pack4(x0, x1, x2, x3, sc, sl[0], sl[1]);
}
// observed dispatch: block b runs on XCD b % 8 -> give each XCD a contiguous range of logical ids (nb a multiple of 8)
__device__ __forceinline__ int xcd_contiguous(int bid, int nb) {
  return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid;
}

// block barrier that orders LDS traffic only: global loads / stores of the wave stay in flight across it
__device__ __forceinline__ void lds_block_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace p2m
