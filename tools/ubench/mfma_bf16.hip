// Sustained bf16 MFMA rates on gfx950 as a function of instruction shape and of the number of independent accumulators
// between two dependent MFMAs (the pattern of the bf16x3 GEMM kernels: six products accumulate into the same tile).
// Every SIMD runs WPS waves; each wave issues `iters` x CH MFMAs, accumulator c = iteration-independent index.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16.hip -o tools/ubench/bin/mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CH>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// same pattern with RANDOM operand bits (full switching activity: the sustained clock under power management, not the issue rate)
__global__ __launch_bounds__(256) void k32_rand(float* out, int iters) {
  bf16x8 a[3], b[3][3];
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (__bf16)(((int)(st >> 9) & 0xffff) * (1.f / 32768.f) - 1.f); };
  for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) a[p][i] = rnd();
  for (int g = 0; g < 3; ++g) for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) b[g][p][i] = rnd();
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  for (int it = 0; it < iters; ++it) {
#define T(ia, ib) _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[g][ib], acc[g], 0, 0, 0);
    T(0, 2) T(2, 0) T(1, 1) T(0, 1) T(1, 0) T(0, 0)
#undef T
  }
  float s = 0;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the operand pattern of the bf16x3 kernels: 3 accumulators, 3 A fragments x 9 B fragments in distinct registers, 18 MFMAs per group
__global__ __launch_bounds__(256) void k32_real(float* out, int iters) {
  bf16x8 a[3], b[3][3];
  for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) a[p][i] = (__bf16)(float)(threadIdx.x + i + p);
  for (int g = 0; g < 3; ++g) for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) b[g][p][i] = (__bf16)(float)(threadIdx.x * 3 + i + p + 5 * g);
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  for (int it = 0; it < iters; ++it) {
#define T(ia, ib) _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[g][ib], acc[g], 0, 0, 0);
    T(0, 2) T(2, 0) T(1, 1) T(0, 1) T(1, 0) T(0, 0)
#undef T
    a[0][0] += (__bf16)1.0f;   // keep the loop body from being hoisted
  }
  float s = 0;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the inner loop of the bf16x3 kernels, element by element: 36 MFMAs per iteration (two groups of 18) on fragments that are
// (MODE & 1) re-read from LDS every iteration (24 ds_read_b128, conflict-free), with (MODE & 2) one s_barrier per iteration;
// NW wavefronts per workgroup, one workgroup per CU (LDS sized to force it when MODE & 4)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void kloop(float* out, int iters) {
  __shared__ u32x4 lds[(MODE & 4) ? 5120 : 2048];   // 80 KB: one workgroup per CU
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += NW * 64) lds[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  bf16x8 a[2][3], b[3][2][3];
  const int l32 = lane & 31, lh = lane >> 5, sw = (l32 >> 2) & 3;
  auto rd = [&](int row0, int kh) { return __builtin_bit_cast(bf16x8, lds[(row0 + l32) * 4 + ((2 * kh + lh) ^ sw)]); };
#define READ_ALL(kh) { for (int p = 0; p < 3; ++p) a[kh][p] = rd(32 * p, kh); for (int g = 0; g < 3; ++g) for (int p = 0; p < 3; ++p) b[g][kh][p] = rd(96 + 32 * (3 * g + p), kh); }
  READ_ALL(0) READ_ALL(1)
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  float vx[8];
  for (int i = 0; i < 8; ++i) vx[i] = 1.0f + 0.001f * (tid + i);
  unsigned vsink = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 8) {   // the staging VALU of the bf16x3 kernels: 8 packed three-way splits (~90 VALU) per 36 MFMAs, independent of them
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        float x = vx[i], y = vx[i + 1];
        for (int r = 0; r < 2; ++r) {
          bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
          unsigned h1 = __builtin_bit_cast(unsigned, p);
          x -= __uint_as_float(h1 << 16); y -= __uint_as_float(h1 & 0xffff0000u);
          p = __builtin_convertvector(f32x2{x, y}, bf16x2);
          unsigned h2 = __builtin_bit_cast(unsigned, p);
          x -= __uint_as_float(h2 << 16); y -= __uint_as_float(h2 & 0xffff0000u);
          p = __builtin_convertvector(f32x2{x, y}, bf16x2);
          vsink ^= h1 ^ h2 ^ __builtin_bit_cast(unsigned, p);
          x = vx[i] * 1.0001f + r; y = vx[i + 1] * 0.9999f + r;
        }
        vx[i] = x; vx[i + 1] = y;
      }
    }
#define T(kh, ia, ib) _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kh][ia], b[g][kh][ib], acc[g], 0, 0, 0);
    if (MODE & 1) READ_ALL(1)
    __builtin_amdgcn_sched_barrier(0);
    T(0, 0, 2) T(0, 2, 0) T(0, 1, 1) T(0, 0, 1) T(0, 1, 0) T(0, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 2) __syncthreads();
    if (MODE & 1) READ_ALL(0)
    __builtin_amdgcn_sched_barrier(0);
    T(1, 0, 2) T(1, 2, 0) T(1, 1, 1) T(1, 0, 1) T(1, 1, 0) T(1, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
#undef T
  }
  float s = 0;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * NW * 64 + tid] = s + (vsink == 12345u ? 1.f : 0.f);
}
// kloop with MODE = 8 (+ LDS reads / barrier), but the staging VALU is INTERLEAVED with the MFMAs in program order
// (sched_group_barrier: 1 MFMA, then VPM VALU, ...) instead of standing in a block in front of them: within a wave an
// independent VALU instruction can issue in the shadow of an executing MFMA only if it FOLLOWS it in program order.
template <int NW, int MODE, int VPM>
__global__ __launch_bounds__(NW * 64) void kloop_il(float* out, int iters) {
  __shared__ u32x4 lds[(MODE & 4) ? 5120 : 2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += NW * 64) lds[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  bf16x8 a[2][3], b[3][2][3];
  const int l32 = lane & 31, lh = lane >> 5, sw = (l32 >> 2) & 3;
  auto rd = [&](int row0, int kh) { return __builtin_bit_cast(bf16x8, lds[(row0 + l32) * 4 + ((2 * kh + lh) ^ sw)]); };
  READ_ALL(0) READ_ALL(1)
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  float vx[8];
  for (int i = 0; i < 8; ++i) vx[i] = 1.0f + 0.001f * (tid + i);
  unsigned vsink = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) READ_ALL(1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      float x = vx[i], y = vx[i + 1];
      for (int r = 0; r < 2; ++r) {
        bf16x2 p = __builtin_convertvector(f32x2{x, y}, bf16x2);
        unsigned h1 = __builtin_bit_cast(unsigned, p);
        x -= __uint_as_float(h1 << 16); y -= __uint_as_float(h1 & 0xffff0000u);
        p = __builtin_convertvector(f32x2{x, y}, bf16x2);
        unsigned h2 = __builtin_bit_cast(unsigned, p);
        x -= __uint_as_float(h2 << 16); y -= __uint_as_float(h2 & 0xffff0000u);
        p = __builtin_convertvector(f32x2{x, y}, bf16x2);
        vsink ^= h1 ^ h2 ^ __builtin_bit_cast(unsigned, p);
        x = vx[i] * 1.0001f + r; y = vx[i + 1] * 0.9999f + r;
      }
      vx[i] = x; vx[i + 1] = y;
    }
#define T(kh, ia, ib) _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kh][ia], b[g][kh][ib], acc[g], 0, 0, 0);
    T(0, 0, 2) T(0, 2, 0) T(0, 1, 1) T(0, 0, 1) T(0, 1, 0) T(0, 0, 0)
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 2) __syncthreads();
    if (MODE & 1) READ_ALL(0)
    __builtin_amdgcn_sched_barrier(0);
    T(1, 0, 2) T(1, 2, 0) T(1, 1, 1) T(1, 0, 1) T(1, 1, 0) T(1, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
#undef T
  }
  float s = 0;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * NW * 64 + tid] = s + (vsink == 12345u ? 1.f : 0.f);
}
template <typename F>
float time_ms(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  const int iters = 4096;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;   // 4 waves per block: one per SIMD; wps blocks per CU
#define RUN16(CH) { float ms = time_ms([&] { hipLaunchKernelGGL(k16<CH>, dim3(blocks), dim3(256), 0, 0, out, iters); }); \
    double n = double(blocks) * 4 * iters * CH; printf("16x16x32 bf16, %d wave(s)/SIMD, %2d independent accumulators: %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", wps, CH, n * 16384 / ms * 1e-9, ms * 1e6 / (double(iters) * CH * wps)); }
#define RUN32(CH) { float ms = time_ms([&] { hipLaunchKernelGGL(k32<CH>, dim3(blocks), dim3(256), 0, 0, out, iters); }); \
    double n = double(blocks) * 4 * iters * CH; printf("32x32x16 bf16, %d wave(s)/SIMD, %2d independent accumulators: %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", wps, CH, n * 32768 / ms * 1e-9, ms * 1e6 / (double(iters) * CH * wps)); }
    RUN16(1) RUN16(2) RUN16(3) RUN16(4) RUN16(8) RUN16(16)
    RUN32(1) RUN32(2) RUN32(3) RUN32(4)
    for (int it2 : {4096, 65536}) { float ms = time_ms([&] { hipLaunchKernelGGL(k32_rand, dim3(blocks), dim3(256), 0, 0, out, it2); });
      double n = double(blocks) * 4 * it2 * 18; printf("32x32x16 bf16, %d wave(s)/SIMD, RANDOM operands, %6.1f ms kernel: %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", wps, ms, n * 32768 / ms * 1e-9, ms * 1e6 / (double(it2) * 18 * wps)); }
    { float ms = time_ms([&] { hipLaunchKernelGGL(k32_real, dim3(blocks), dim3(256), 0, 0, out, iters); });
      double n = double(blocks) * 4 * iters * 18; printf("32x32x16 bf16, %d wave(s)/SIMD, bf16x3 operand pattern (12 distinct fragments, 3 accumulators): %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", wps, n * 32768 / ms * 1e-9, ms * 1e6 / (double(iters) * 18 * wps)); }
  }
#define RUNL(NW, MODE, BLK) { float ms = time_ms([&] { hipLaunchKernelGGL((kloop<NW, MODE>), dim3(BLK), dim3(NW * 64), 0, 0, out, 2048); }); \
    printf("loop of 36 MFMAs, %d waves / workgroup, %d workgroups%s%s%s%s: %.1f ns per MFMA per SIMD\n", NW, BLK, (MODE & 1) ? ", 24 LDS fragment reads" : "", (MODE & 2) ? ", 1 barrier" : "", (MODE & 4) ? ", 1 workgroup / CU" : "", (MODE & 8) ? ", + ~90 independent staging VALU" : "", ms * 1e6 / (2048.0 * 36 * ((MODE & 4) ? NW / 4.0 : (double(BLK) * NW / 4 / 256)))); }
  RUNL(8, 4, 256) RUNL(8, 5, 256) RUNL(8, 6, 256) RUNL(8, 7, 256)
  RUNL(4, 0, 512) RUNL(4, 1, 512) RUNL(4, 2, 512) RUNL(4, 3, 512)
  RUNL(8, 12, 256) RUNL(8, 15, 256) RUNL(4, 8, 512) RUNL(4, 11, 512)
#define RUNI(NW, MODE, VPM, BLK) { float ms = time_ms([&] { hipLaunchKernelGGL((kloop_il<NW, MODE, VPM>), dim3(BLK), dim3(NW * 64), 0, 0, out, 2048); }); \
    printf("loop of 36 MFMAs, %d waves / workgroup, %d workgroups%s%s%s, + ~90 staging VALU INTERLEAVED %d per MFMA of the first group: %.1f ns per MFMA per SIMD\n", NW, BLK, (MODE & 1) ? ", 24 LDS fragment reads" : "", (MODE & 2) ? ", 1 barrier" : "", (MODE & 4) ? ", 1 workgroup / CU" : "", VPM, ms * 1e6 / (2048.0 * 36 * ((MODE & 4) ? NW / 4.0 : (double(BLK) * NW / 4 / 256)))); }
  RUNI(8, 12, 3, 256) RUNI(8, 12, 5, 256) RUNI(8, 12, 7, 256) RUNI(8, 15, 5, 256) RUNI(4, 8, 5, 512) RUNI(4, 11, 5, 512)
  return 0;
}
