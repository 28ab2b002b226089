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

// ---- the same tile with the score GEMM on v_mfma_f32_16x16x32_bf16: the K = 4 contraction of fp32 operands becomes 24 of
// the 32 K slots - the six exact bf16 x bf16 products (w1 x1, w1 x2, w1 x3, w2 x1, w2 x2, w3 x1) of every feature laid side
// by side - so ONE bf16 MFMA (16 cycles) replaces the fp32 MFMA (32 cycles) at fp32 accuracy.  Per tile every lane splits
// the four features of its edge (two packed splits) and picks the two 4-slot groups of its K range; the weight operand is
// pre-arranged per column tile: in registers (WLDS = false, 64 VGPRs) or read from LDS per tile (WLDS = true).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split3 { unsigned h1, h2, h3; };
__device__ __forceinline__ Split3 split_pair(float x, float y) {
  Split3 s;
  bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
  s.h1 = __builtin_bit_cast(unsigned, p);
  x -= __uint_as_float(s.h1 << 16); y -= __uint_as_float(s.h1 & 0xffff0000u);
  p = __builtin_convertvector(f32x2{x, y}, bf16x2);
  s.h2 = __builtin_bit_cast(unsigned, p);
  x -= __uint_as_float(s.h2 << 16); y -= __uint_as_float(s.h2 & 0xffff0000u);
  p = __builtin_convertvector(f32x2{x, y}, bf16x2);
  s.h3 = __builtin_bit_cast(unsigned, p);
  return s;
}

template <int DEPTH, bool WLDS>
__global__ __launch_bounds__(256, 2) void k_tile_bf16(const float* __restrict__ w, float* __restrict__ out, int tiles) {
  __shared__ u32x4 sW[CT * 64];
  const int lane = threadIdx.x & 63, g = lane >> 4;
  float att[CT][4];
  f32x4 cinit[CT];
  u32x4 Wa[WLDS ? 1 : CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const u32x4 v = u32x4{__float_as_uint(w[(ct * 64 + lane) & 1023]) & 0xffff0000u, 0x3f803f80u, 0x3e803e80u, 0x3d803d80u};
    if (WLDS) { if (threadIdx.x < 64) sW[ct * 64 + lane] = v; } else Wa[ct] = v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      att[ct][r] = w[(ct * 4 + r + lane) & 1023] * 0.01f;
      cinit[ct][r] = w[(ct * 4 + r + 2 * lane) & 1023];
    }
  }
  __syncthreads();
  float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  float4 xe = make_float4(w[lane], w[lane + 1], w[lane + 2], w[lane + 3]);
  for (int t = 0; t < tiles; ++t) {
    // B operand: k slots 8g..8g+7 of edge j = two of the groups x1 | x2 | x3 (each = the 4 features as 2 packed words)
    const Split3 p01 = split_pair(xe.x, xe.y), p23 = split_pair(xe.z, xe.w);
    u32x4 xb;
    xb[0] = g == 0 ? p01.h1 : g == 1 ? p01.h3 : g == 2 ? p01.h2 : 0u;
    xb[1] = g == 0 ? p23.h1 : g == 1 ? p23.h3 : g == 2 ? p23.h2 : 0u;
    xb[2] = g == 0 ? p01.h2 : g == 3 ? 0u : p01.h1;
    xb[3] = g == 0 ? p23.h2 : g == 3 ? 0u : p23.h1;
    const bf16x8 xB = __builtin_bit_cast(bf16x8, xb);
    float pe[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) { pe[k][0] = att[k][0] * xe.x; pe[k][1] = 0.f; }
    f32x4 z[CT];
#define WOP(ct) __builtin_bit_cast(bf16x8, WLDS ? sW[(ct) * 64 + lane] : Wa[WLDS ? 0 : (ct)])
#pragma unroll
    for (int ct = 0; ct < DEPTH && ct < CT; ++ct) z[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WOP(ct), xB, cinit[ct], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int k = ct / 4;
      pe[k][0] = fmaf(att[ct][0], fabsf(z[ct][0]), pe[k][0]);
      pe[k][1] = fmaf(att[ct][1], fabsf(z[ct][1]), pe[k][1]);
      pe[k][0] = fmaf(att[ct][2], fabsf(z[ct][2]), pe[k][0]);
      pe[k][1] = fmaf(att[ct][3], fabsf(z[ct][3]), pe[k][1]);
      __builtin_amdgcn_sched_barrier(0);
      if (ct + DEPTH < CT) {
        z[ct + DEPTH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WOP(ct + DEPTH), xB, cinit[ct + DEPTH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef WOP
    float e = reduce_heads(pe[0][0] + pe[0][1], pe[1][0] + pe[1][1], pe[2][0] + pe[2][1], pe[3][0] + pe[3][1]);
    const float mn = fmaxf(m, e);
    const float sc = __builtin_amdgcn_exp2f(m - mn);
    const float p = __builtin_amdgcn_exp2f(e - mn);
    den = fmaf(den, sc, p);
    s0 = fmaf(s0, sc, p * xe.x);
    s1 = fmaf(s1, sc, p * xe.y);
    s2 = fmaf(s2, sc, p * xe.z);
    s3 = fmaf(s3, sc, p * xe.w);
    m = mn;
    xe.x = xe.x * 0.999f + 0.001f * p; xe.y += 0.001f; xe.z -= 0.001f; xe.w = xe.w * 0.999f;
  }
  if (den + s0 + s1 + s2 + s3 == 12345.f) out[0] = m;
}

// ---- round 3: the same bf16x3-in-K tile with a LEAN operand layout: lane (j, g) owns ONLY feature g of edge j.  K group g
// (8 slots) holds the six products of feature g: A = (w1 w1 | w2 w2 | w1 w3 | 0 0), B = (x1 x2 | x1 x2 | x3 x1 | 0 0), so a
// lane splits ONE value per tile (3 cvt + 2 and + 2 sub + 2 bfi = 9 VALU instead of two packed splits + 12 selects).
// NOFMA = true: the |z| FMAs do not read the MFMA results (upper bound of what MFMA / VALU overlap can give).
template <int DEPTH, bool NOFMA>
__global__ __launch_bounds__(256, 2) void k_tile_bf16_lean(const float* __restrict__ w, float* __restrict__ out, int tiles) {
  const int lane = threadIdx.x & 63;
  float att[CT][4];
  f32x4 cinit[CT];
  u32x4 Wa[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float wv = w[(ct * 64 + lane) & 1023];
    const Split3 sp = split_pair(wv, wv);
    Wa[ct] = u32x4{sp.h1, sp.h2, (sp.h1 & 0xffffu) | (sp.h3 & 0xffff0000u), 0u};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      att[ct][r] = w[(ct * 4 + r + lane) & 1023] * 0.01f;
      cinit[ct][r] = w[(ct * 4 + r + 2 * lane) & 1023];
    }
  }
  float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  float4 xe = make_float4(w[lane], w[lane + 1], w[lane + 2], w[lane + 3]);
  float xg = w[lane + 7];       // the lane's own feature of its edge
  for (int t = 0; t < tiles; ++t) {
    // B operand of the lane's K group: one three-way split
    bf16x2 p = __builtin_convertvector(f32x2{xg, xg}, bf16x2);
    const unsigned h1 = __builtin_bit_cast(unsigned, p);
    const float r1 = xg - __uint_as_float(h1 & 0xffff0000u);
    p = __builtin_convertvector(f32x2{r1, r1}, bf16x2);
    const unsigned h2 = __builtin_bit_cast(unsigned, p);
    const float r2 = r1 - __uint_as_float(h2 & 0xffff0000u);
    p = __builtin_convertvector(f32x2{r2, r2}, bf16x2);
    const unsigned h3 = __builtin_bit_cast(unsigned, p);
    const unsigned x12 = (h1 & 0xffffu) | (h2 & 0xffff0000u);
    const u32x4 xb = u32x4{x12, x12, (h3 & 0xffffu) | (h1 & 0xffff0000u), 0u};
    const bf16x8 xB = __builtin_bit_cast(bf16x8, xb);
    float pe[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) { pe[k][0] = att[k][0] * xg; pe[k][1] = 0.f; }
    f32x4 z[CT];
#pragma unroll
    for (int ct = 0; ct < DEPTH && ct < CT; ++ct)
      z[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wa[ct]), xB, cinit[ct], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int k = ct / 4;
      if (!NOFMA) {
        pe[k][0] = fmaf(att[ct][0], fabsf(z[ct][0]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][1], fabsf(z[ct][1]), pe[k][1]);
        pe[k][0] = fmaf(att[ct][2], fabsf(z[ct][2]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][3], fabsf(z[ct][3]), pe[k][1]);
      } else {
        pe[k][0] = fmaf(att[ct][0], fabsf(att[ct][1]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][1], fabsf(att[ct][2]), pe[k][1]);
        pe[k][0] = fmaf(att[ct][2], fabsf(att[ct][3]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][3], fabsf(att[ct][0]), pe[k][1]);
        if (ct == CT - 1) {
          _Pragma("unroll") for (int c2 = 0; c2 < CT; ++c2) pe[0][0] += z[c2][0];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ct + DEPTH < CT) {
        z[ct + DEPTH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wa[ct + DEPTH]), xB, cinit[ct + DEPTH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float e = reduce_heads(pe[0][0] + pe[0][1], pe[1][0] + pe[1][1], pe[2][0] + pe[2][1], pe[3][0] + pe[3][1]);
    const float mn = fmaxf(m, e);
    const float sc = __builtin_amdgcn_exp2f(m - mn);
    const float pp = __builtin_amdgcn_exp2f(e - mn);
    den = fmaf(den, sc, pp);
    s0 = fmaf(s0, sc, pp * xe.x);
    s1 = fmaf(s1, sc, pp * xe.y);
    s2 = fmaf(s2, sc, pp * xe.z);
    s3 = fmaf(s3, sc, pp * xe.w);
    m = mn;
    xg = xg * 0.999f + 0.001f * pp;
    xe.x = xe.x * 0.999f + 0.001f * pp; xe.y += 0.001f; xe.z -= 0.001f; xe.w = xe.w * 0.999f;
  }
  if (den + s0 + s1 + s2 + s3 == 12345.f) out[0] = m;
}

template <int DEPTH, bool NOFMA>
void run_bf16_lean(const float* w, float* out, int wps, const char* name) {
  const int tiles = 4000;
  const int blocks = 256 * wps;
  auto launch = [&] { hipLaunchKernelGGL((k_tile_bf16_lean<DEPTH, NOFMA>), dim3(blocks), dim3(256), 0, 0, w, out, tiles); };
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
  const double ns = us * 1e3 / (double(tiles) * wps);
  printf("%-48s waves/SIMD=%d  %8.1f us  %7.1f ns per tile per SIMD  (= %6.0f cyc @2.0GHz)\n", name, wps, us, ns, ns * 2.0);
}

template <int DEPTH, bool WLDS>
void run_bf16(const float* w, float* out, int wps, const char* name) {
  const int tiles = 4000;
  const int blocks = 256 * wps;
  auto launch = [&] { hipLaunchKernelGGL((k_tile_bf16<DEPTH, WLDS>), dim3(blocks), dim3(256), 0, 0, w, out, tiles); };
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
  const double ns = us * 1e3 / (double(tiles) * wps);
  printf("%-48s waves/SIMD=%d  %8.1f us  %7.1f ns per tile per SIMD  (= %6.0f cyc @2.0GHz)  bf16 MFMA-only floor 256 cyc\n", name, wps, us, ns, ns * 2.0);
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
    run_bf16_lean<1, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 1");
    run_bf16_lean<2, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 2");
    run_bf16_lean<4, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 4");
    run_bf16_lean<3, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 3");
    run_bf16_lean<6, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 6");
    run_bf16_lean<8, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 8");
    run_bf16_lean<16, false>(w, out, wps, "bf16x3-in-K LEAN (lane = one feature), depth 16");
    run_bf16_lean<2, true>(w, out, wps, "bf16x3-in-K LEAN, FMAs independent of z, depth 2");
    run_bf16<1, false>(w, out, wps, "bf16x3-in-K score MFMA, depth 1, W in registers");
    run_bf16<2, false>(w, out, wps, "bf16x3-in-K score MFMA, depth 2, W in registers");
    run_bf16<4, false>(w, out, wps, "bf16x3-in-K score MFMA, depth 4, W in registers");
    run_bf16<2, true>(w, out, wps, "bf16x3-in-K score MFMA, depth 2, W from LDS");
    run_bf16<4, true>(w, out, wps, "bf16x3-in-K score MFMA, depth 4, W from LDS");
  }
  return 0;
}
