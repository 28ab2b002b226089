// Which instruction order lets v_mfma_f32_16x16x4_f32 and the dependent |z| FMAs of the K1 row tile overlap?
// Body = K1's tile: 16 MFMA (C operand = per-destination registers) + 64 |z| FMAs + head reduction + softmax update.
// DEPTH = how many MFMAs are issued ahead of the FMAs that consume them (1 = what hipcc emits for the plain loop).
// Wall-clock (events), whole chip, WPS waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CT = 16;

__device__ __forceinline__ float reduce_heads(float pe0, float pe1, float pe2, float pe3) {
  auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe0), __float_as_uint(pe2), false, false);
  const float a = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);
  auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe1), __float_as_uint(pe3), false, false);
  const float b = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

template <int DEPTH, bool PIN, int EXTRA, int MODE = 0>
__global__ __launch_bounds__(256, 2) void k_tile(const float* __restrict__ w, float* __restrict__ out, int tiles) {
  const int lane = threadIdx.x & 63;
  float Wa[CT], att[CT][4];
  f32x4 cinit[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    Wa[ct] = w[(ct * 64 + lane) & 1023];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      att[ct][r] = w[(ct * 4 + r + lane) & 1023] * 0.01f;
      cinit[ct][r] = w[(ct * 4 + r + 2 * lane) & 1023];
    }
  }
  float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  float xB = w[lane];
  for (int t = 0; t < tiles; ++t) {
    float pe[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) { pe[k][0] = att[k][0] * xB; pe[k][1] = 0.f; }
    f32x4 z[CT];
#pragma unroll
    for (int ct = 0; ct < DEPTH && ct < CT; ++ct) z[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[ct], xB, cinit[ct], 0, 0, 0);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int k = ct / 4;
      if (MODE == 0 || MODE == 2) {
        pe[k][0] = fmaf(att[ct][0], fabsf(z[ct][0]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][1], fabsf(z[ct][1]), pe[k][1]);
        pe[k][0] = fmaf(att[ct][2], fabsf(z[ct][2]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][3], fabsf(z[ct][3]), pe[k][1]);
      } else if (MODE == 1) {   // same FMAs, operands independent of the MFMA results
        pe[k][0] = fmaf(att[ct][0], fabsf(att[ct][1]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][1], fabsf(att[ct][2]), pe[k][1]);
        pe[k][0] = fmaf(att[ct][2], fabsf(att[ct][3]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][3], fabsf(att[ct][0]), pe[k][1]);
        if (ct == CT - 1) {
          _Pragma("unroll") for (int c2 = 0; c2 < CT; ++c2) pe[0][0] += z[c2][0];
        }
      } else {
        if (ct == CT - 1) {
          _Pragma("unroll") for (int c2 = 0; c2 < CT; ++c2) pe[0][0] += z[c2][0] + z[c2][1] + z[c2][2] + z[c2][3];
        }
      }
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      if (ct + DEPTH < CT) {
        if (MODE == 2) { z[ct + DEPTH] = cinit[ct + DEPTH]; z[ct + DEPTH] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[ct + DEPTH], xB, z[ct + DEPTH], 0, 0, 0); }
        else
        z[ct + DEPTH] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[ct + DEPTH], xB, cinit[ct + DEPTH], 0, 0, 0);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
      }
    }
    float e = reduce_heads(pe[0][0] + pe[0][1], pe[1][0] + pe[1][1], pe[2][0] + pe[2][1], pe[3][0] + pe[3][1]);
#pragma unroll
    for (int x = 0; x < EXTRA; ++x) e = fmaf(e, 1.0001f, 0.001f);   // stands for bookkeeping VALU
    const float mn = fmaxf(m, e);
    const float sc = __builtin_amdgcn_exp2f(m - mn);
    const float p = __builtin_amdgcn_exp2f(e - mn);
    den = fmaf(den, sc, p);
    s0 = fmaf(s0, sc, p * xB);
    s1 = fmaf(s1, sc, p * Wa[0]);
    s2 = fmaf(s2, sc, p * Wa[1]);
    s3 = fmaf(s3, sc, p * Wa[2]);
    m = mn;
    xB = xB * 0.999f + 0.001f * p;   // next tile's input depends on nothing slow
  }
  if (den + s0 + s1 + s2 + s3 == 12345.f) out[0] = m;
}

template <int DEPTH, bool PIN, int EXTRA, int MODE = 0>
void run(const float* w, float* out, int wps, const char* name) {
  const int tiles = 4000;
  const int blocks = 256 * wps;   // 256 threads = 1 wave per SIMD per block
  auto launch = [&] { hipLaunchKernelGGL((k_tile<DEPTH, PIN, EXTRA, MODE>), dim3(blocks), dim3(256), 0, 0, w, out, tiles); };
  launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / 3;
  const double ns_per_tile_simd = us * 1e3 / (double(tiles) * wps);   // per SIMD: wps waves x tiles
  printf("%-40s waves/SIMD=%d  %8.1f us  %7.1f ns per tile per SIMD  (= %6.0f cyc @2.0GHz)  MFMA-only floor 512 cyc\n", name,
         wps, us, ns_per_tile_simd, ns_per_tile_simd * 2.0);
}

int main() {
  float *w, *out;
  hipMalloc(&w, 4096);
  hipMalloc(&out, 64);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.01f * ((i * 37) % 101) - 0.5f;
  hipMemcpy(w, h, 4096, hipMemcpyHostToDevice);
  for (int wps : {1, 2}) {
    run<1, false, 0>(w, out, wps, "depth 1, compiler order");
    run<1, true, 0>(w, out, wps, "depth 1, pinned");
    run<2, true, 0>(w, out, wps, "depth 2, pinned");
    run<4, true, 0>(w, out, wps, "depth 4, pinned");
    run<8, true, 0>(w, out, wps, "depth 8, pinned");
    run<16, true, 0>(w, out, wps, "depth 16 (all MFMAs first), pinned");
    run<16, false, 0>(w, out, wps, "depth 16, compiler order");
    run<4, true, 40>(w, out, wps, "depth 4, pinned, +40 bookkeeping VALU");
    run<1, false, 40>(w, out, wps, "depth 1, compiler, +40 bookkeeping VALU");
    run<4, true, 0, 1>(w, out, wps, "depth 4, FMAs independent of z");
    run<16, false, 0, 1>(w, out, wps, "compiler order, FMAs independent of z");
    run<16, false, 0, 3>(w, out, wps, "16 MFMA + sum(z) only (no |z| FMAs)");
    run<16, true, 0, 3>(w, out, wps, "16 MFMA + sum(z) only, pinned");
  }
  return 0;
}
