// f16 MFMA on gfx950 for the two-term split ("f16x2": hi + lo of an exactly power-of-two-scaled fp32 value, three products per fp32
// product instead of bf16x3's six): (1) sustained rate of v_mfma_f32_32x32x16_f16 against v_mfma_f32_32x32x16_bf16 in the operand /
// accumulator pattern of the GRU cell kernel, with random operand bits (the part runs MFMA-dense loops at its power limit);
// (2) how the matrix core treats SUBNORMAL f16 inputs (the lo term of a small element) and whether products are exact in fp32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16.hip -o tools/ubench/bin/mfma_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool F16, int TERMS>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (((int)(st >> 9) & 0xffff) * (1.f / 32768.f) - 1.f); };
  f16x8 ah[3], bh[3][3];
  bf16x8 ab[3], bb[3][3];
  for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) { float v = rnd(); ah[p][i] = (_Float16)v; ab[p][i] = (__bf16)v; }
  for (int g = 0; g < 3; ++g) for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) { float v = rnd(); bh[g][p][i] = (_Float16)v; bb[g][p][i] = (__bf16)v; }
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  for (int it = 0; it < iters; ++it) {
#define T(ia, ib) _Pragma("unroll") for (int g = 0; g < 3; ++g) { if (F16) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ia], bh[g][ib], acc[g], 0, 0, 0); else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[ia], bb[g][ib], acc[g], 0, 0, 0); }
    if (TERMS == 6) { T(0, 2) T(2, 0) T(1, 1) T(0, 1) T(1, 0) T(0, 0) }
    else { T(0, 1) T(1, 0) T(0, 0) }
#undef T
  }
  float s = 0;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// one 32x32x16 product with chosen operand values: A row r, k -> a(r, k); B col c, k -> b(c, k); lane l holds row/col l % 32, k = 8 (l / 32) .. + 7
__global__ void probe(const float* av, const float* bv, float* out) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av[(l & 31) * 16 + 8 * (l >> 5) + i]; b[i] = (_Float16)bv[(l & 31) * 16 + 8 * (l >> 5) + i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[(8 * (i >> 2) + 4 * (l >> 5) + (i & 3)) * 32 + (l & 31)] = c[i];   // D: lane = column l % 32, reg i = row
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  auto run = [&](auto kern, int terms, const char* name) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mf = 256.0 * 8 * iters * terms * 3;                 // MFMAs
      printf("%-34s %8.1f us  %6.2f ns per MFMA per SIMD  %7.1f TFLOP/s\n", name, ms * 1e3, ms * 1e6 / (iters * terms * 3 * 2.0), mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
    }
  };
  run(rate<false, 6>, 6, "bf16 32x32x16, six terms (bf16x3)");
  run(rate<true, 6>, 6, "f16  32x32x16, six terms");
  run(rate<true, 3>, 3, "f16  32x32x16, three terms (f16x2)");
  run(rate<false, 3>, 3, "bf16 32x32x16, three terms");
  // ---- subnormal / exactness probe
  float ha[32 * 16] = {0}, hb[32 * 16] = {0}, ho[32 * 32];
  // row 0: a = 2^-20 (f16 subnormal) at k = 0, b(col 0) = 2^10 -> 2^-10 if subnormal inputs are honoured, 0 if flushed
  ha[0 * 16 + 0] = ldexpf(1.f, -20); hb[0 * 16 + 0] = ldexpf(1.f, 10);
  // row 1: a = 2^-24 (smallest f16 subnormal) x b(col 1) = 1
  ha[1 * 16 + 0] = ldexpf(1.f, -24); hb[1 * 16 + 0] = 1.f;
  // row 2 x col 2: (1 + 2^-10) x (1 + 2^-10) = 1 + 2^-9 + 2^-20: exact in fp32 (21 bits)
  ha[2 * 16 + 0] = 1.f + ldexpf(1.f, -10); hb[2 * 16 + 0] = 1.f + ldexpf(1.f, -10);
  // row 3 x col 3: 2047 x 2047 + 2^-14 x 2^-14 (k = 0, 1): 4190209 + 2^-28 -> fp32 rounds to 4190209 (exact sum needs 51 bits)
  ha[3 * 16 + 0] = 2047.f; hb[3 * 16 + 0] = 2047.f; ha[3 * 16 + 1] = ldexpf(1.f, -14); hb[3 * 16 + 1] = ldexpf(1.f, -14);
  // row 4 x col 4: 16 terms of (1 + 2^-10)^2: 16 + 2^-5 + 2^-16 exact
  for (int k = 0; k < 16; ++k) { ha[4 * 16 + k] = 1.f + ldexpf(1.f, -10); hb[4 * 16 + k] = 1.f + ldexpf(1.f, -10); }
  // row 5 x col 5: 65504 x 65504 (largest finite f16): 4290774016 exact in fp32
  ha[5 * 16 + 0] = 65504.f; hb[5 * 16 + 0] = 65504.f;
  // row 6 x col 6: cancellation inside one instruction: 2048 x 2048 - 2048 x 2048 + 2^-12 x 2^-12 = 2^-24 if the internal sum is wide
  ha[6 * 16 + 0] = 2048.f; hb[6 * 16 + 0] = 2048.f; ha[6 * 16 + 1] = -2048.f; hb[6 * 16 + 1] = 2048.f; ha[6 * 16 + 2] = ldexpf(1.f, -12); hb[6 * 16 + 2] = ldexpf(1.f, -12);
  float *da, *db, *dout;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(ho));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
  hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  printf("subnormal a = 2^-20 x 2^10:          got %.10e  (honoured: %.10e, flushed: 0)\n", ho[0 * 32 + 0], ldexp(1.0, -10));
  printf("subnormal a = 2^-24 x 1:             got %.10e  (honoured: %.10e)\n", ho[1 * 32 + 1], ldexp(1.0, -24));
  printf("(1 + 2^-10)^2:                       got %.10e  exact %.10e\n", ho[2 * 32 + 2], (1 + ldexp(1.0, -10)) * (1 + ldexp(1.0, -10)));
  printf("2047^2 + 2^-28:                      got %.10e  fp32(exact) %.10e\n", ho[3 * 32 + 3], (double)(float)(2047.0 * 2047.0 + ldexp(1.0, -28)));
  printf("16 x (1 + 2^-10)^2:                  got %.10e  exact %.10e\n", ho[4 * 32 + 4], 16 * (1 + ldexp(1.0, -10)) * (1 + ldexp(1.0, -10)));
  printf("65504^2:                             got %.10e  exact %.10e\n", ho[5 * 32 + 5], 65504.0 * 65504.0);
  printf("2048^2 - 2048^2 + 2^-24:             got %.10e  wide internal sum: %.10e\n", ho[6 * 32 + 6], ldexp(1.0, -24));
  return 0;
}
