// fp32 products of a K = 4 (or 2 + 2) contraction on the bf16 matrix cores, shared by the K1 forward (csrc/gatv2_hetero.hip) and
// the matrix-core K1 backward (csrc/gatv2.hip): operand layouts of ONE v_mfma_f32_16x16x32_bf16 per 16 x 16 tile.
#ifndef UAVGNN_K1_X3_H
#define UAVGNN_K1_X3_H
#include "common.h"

namespace uavgnn {
namespace {

// The K = 4 contraction of a tile is laid out over the K = 32 of ONE v_mfma_f32_16x16x32_bf16: K group g (8 slots, = lane
// group g on both operands) holds the six bf16 x bf16 products of feature g -
//     A = (w1 w1 | w2 w2 | w1 w3 | c c')      B = (x1 x2 | x1 x2 | x3 x1 | d d')      w = w1 + w2 + w3, x = x1 + x2 + x3 exactly
// and the last word of a group carries bias terms against 1 (or against a 0 / 1 flag).
typedef __bf16 k1_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 k1_bf16x8 __attribute__((ext_vector_type(8)));
typedef float k1_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned k1_u32x4 __attribute__((ext_vector_type(4)));
struct K1Split { unsigned h1, h2, h3; };   // (t1 | t1 << 16) of the three terms
__device__ __forceinline__ K1Split k1_split(float x) {
  K1Split s;
  k1_bf16x2 p = __builtin_convertvector(k1_f32x2{x, x}, k1_bf16x2);
  s.h1 = __builtin_bit_cast(unsigned, p);
  const float r1 = x - __uint_as_float(s.h1 & 0xffff0000u);
  p = __builtin_convertvector(k1_f32x2{r1, r1}, k1_bf16x2);
  s.h2 = __builtin_bit_cast(unsigned, p);
  const float r2 = r1 - __uint_as_float(s.h2 & 0xffff0000u);
  p = __builtin_convertvector(k1_f32x2{r2, r2}, k1_bf16x2);
  s.h3 = __builtin_bit_cast(unsigned, p);
  return s;
}
// A operand of one (row, K group): weight w, and in the last word the bias triple's first two terms (pair) or its third
__device__ __forceinline__ k1_u32x4 k1_a_operand(float w, float bias, int bias_part) {   // bias_part: 0 none, 1 (b1 b2), 2 (b3 0)
  const K1Split s = k1_split(w), b = k1_split(bias);
  const unsigned last = bias_part == 1 ? ((b.h1 & 0xffffu) | (b.h2 & 0xffff0000u)) : bias_part == 2 ? (b.h3 & 0xffffu) : 0u;
  return k1_u32x4{s.h1, s.h2, (s.h1 & 0xffffu) | (s.h3 & 0xffff0000u), last};
}
__device__ __forceinline__ k1_bf16x8 k1_b_operand(float x, unsigned last_word) {   // (x1 x2 | x1 x2 | x3 x1 | last)
  const K1Split s = k1_split(x);
  const unsigned x12 = (s.h1 & 0xffffu) | (s.h2 & 0xffff0000u);
  return __builtin_bit_cast(k1_bf16x8, k1_u32x4{x12, x12, (s.h3 & 0xffffu) | (s.h1 & 0xffff0000u), last_word});
}
#define K1_MFMA(WA_ct, XBOP, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k1_bf16x8, WA_ct), XBOP, C, 0, 0, 0)

}  // namespace
}  // namespace uavgnn
#endif  // UAVGNN_K1_X3_H
