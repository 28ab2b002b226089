// fp32 GEMM arithmetic on the bf16 matrix cores of gfx950 ("bf16x3"): shared device helpers.
//
// fp32 MFMA on gfx950 issues at the fp32 VECTOR rate - 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md, peak table) - so
// the GEMM-shaped parts of the path (GRU cell, nn.Linear layers) compute their fp32 products on v_mfma_f32_16x16x32_bf16:
//   * every fp32 operand is split EXACTLY into three bf16 terms, a = a1 + a2 + a3 (round-to-nearest residuals: 8 + 8 + 8
//     significand bits cover the 24 of fp32; bf16 has fp32's exponent range, so no scaling is needed);
//   * a*b is accumulated as the six bf16 x bf16 products of weight >= 2^-16: a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 - each
//     EXACT in the fp32 accumulator (8 x 8 significand bits); the three dropped products are <= 2^-23 |a b| together, one
//     fp32 rounding of the product the fp32 pipeline would have made;
//   * accumulation is fp32 in the MFMA accumulator, smallest terms first.
// Measured against an fp64 reference (tools/ubench/gemm_bf16x3.hip, profiles/r02_ubench_gemm_bf16x3.txt): error / sum|a b|
// max 1.4e-7, mean 1.6e-8 - BELOW rocBLAS sgemm on the same data (2.2e-7 / 2.0e-8) - at 1.4-1.8x its rate.
#pragma once
#include <hip/hip_runtime.h>

namespace uavgnn {
namespace x3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Split3 {
  unsigned h1, h2, h3;   // two packed bf16 each: low half = first element
};

// (x, y) -> three packed bf16 pairs with x = x1 + x2 + x3 exactly (v_cvt_pk_bf16_f32 rounds to nearest even)
__device__ __forceinline__ Split3 split_pair(float x, float y) {
  Split3 s;
  bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
  s.h1 = __builtin_bit_cast(unsigned, p);
  x -= __uint_as_float(s.h1 << 16);
  y -= __uint_as_float(s.h1 & 0xffff0000u);
  p = __builtin_convertvector(f32x2{x, y}, bf16x2);
  s.h2 = __builtin_bit_cast(unsigned, p);
  x -= __uint_as_float(s.h2 << 16);
  y -= __uint_as_float(s.h2 & 0xffff0000u);
  p = __builtin_convertvector(f32x2{x, y}, bf16x2);
  s.h3 = __builtin_bit_cast(unsigned, p);
  return s;
}

// LDS tiles hold one K-slice of 32 bf16 per row = four 16-byte chunks, no padding; chunk c of row r lives at r * 4 +
// (c ^ swz(r)).  ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... :
// MI355X_MICROARCH.md, LDS table), i.e. a fragment read (lane = 16 g + j -> row j, chunk g) puts rows {0-3,12-15} with one g
// and rows {4-11} with g ^ 1 into one group; the swizzle makes those 16 chunks distinct mod 256 B for all four groups.
__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }

// stage four consecutive k of one row (16-byte global load) as three 8-byte bf16 groups, one per split plane
__device__ __forceinline__ void stage4(unsigned short* p, int plane_stride, float4 v) {
  const Split3 a = split_pair(v.x, v.y), b = split_pair(v.z, v.w);
  *reinterpret_cast<u32x2*>(p) = u32x2{a.h1, b.h1};
  *reinterpret_cast<u32x2*>(p + plane_stride) = u32x2{a.h2, b.h2};
  *reinterpret_cast<u32x2*>(p + 2 * plane_stride) = u32x2{a.h3, b.h3};
}

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 32x32x16 fragments (lane -> row lane % 32, chunk 2 kh + lane / 32 of the 32-wide slice): the swizzle that keeps the four
// 16-lane groups of a ds_read_b128 on distinct 16-byte bank groups is the row's 4-row block index
__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

}  // namespace x3
}  // namespace uavgnn
