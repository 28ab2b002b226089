// Micro-benchmarks that decide the K1 design (gfx950):
//   1. does v_mfma_f32_16x16x4_f32 overlap with VALU fp32 work of the SAME wave / of the partner wave on the same SIMD?
//   2. issue cost of v_fma_f32 vs v_pk_fma_f32 (per wave-instruction)
//   3. kernel start-up: empty kernel vs 2048 waves loading ~130 per-lane constants
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o gpurun_out/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM, int NV, bool PK>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a, float b) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{v[i], v[i] + 1.f}; }
  const f32x2 pa{a, a}, pb{b, b};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
      if (NM > 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (PK) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k & 7]) : "v"(pa), "v"(pb));
        } else {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
        }
      }
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = static_cast<float>(t1 - t0);
}

// VALU consuming the MFMA result of the previous MFMA (dependent, like the |z| FMAs of K1)
template <int NV>
__global__ __launch_bounds__(256) void k_dep(float* out, int iters, float a, float b) {
  f32x4 c{0.1f, 0.2f, 0.3f, 0.4f};
  float pe0 = 0.f, pe1 = 0.f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const f32x4 z = __builtin_amdgcn_mfma_f32_16x16x4f32(a + m, b, c, 0, 0, 0);
      pe0 = fmaf(a, fabsf(z[0]), pe0);
      pe1 = fmaf(a, fabsf(z[1]), pe1);
      pe0 = fmaf(a, fabsf(z[2]), pe0);
      pe1 = fmaf(a, fabsf(z[3]), pe1);
#pragma unroll
      for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(pe0) : "v"(a), "v"(b));
    }
  }
  long long t1 = clock64();
  if (pe0 + pe1 == 123.456f) out[0] = pe0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = static_cast<float>(t1 - t0);
}

__global__ void k_empty(float* out) {
  if (out == nullptr) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_consts(const float* __restrict__ w, float* out, int n) {
  float acc = 0.f;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 130; ++i) acc += w[(lane * 2 + i * 7) % n];
  if (acc == 123.456f) out[0] = acc;
}

template <typename F>
float time_us(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

template <int NM, int NV, bool PK>
void run_mix(float* d_out, int blocks, int threads, const char* name) {
  const int iters = 2000;
  hipLaunchKernelGGL((k_mix<NM, NV, PK>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  float h[2];
  hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
  const double cyc = h[1] / iters;
  printf("%-44s blocks=%4d thr=%3d  cycles/iter %8.1f  (per MFMA %6.1f, per VALU %5.2f)\n", name, blocks, threads, cyc,
         NM ? cyc / NM : 0.0, NV ? cyc / ((NM > 0 ? NM : 1) * NV) : 0.0);
}

int main() {
  float* d_out; hipMalloc(&d_out, 1024);
  float* d_w; hipMalloc(&d_w, 4096 * 4); hipMemset(d_w, 0, 4096 * 4);
  printf("== clock64 ticks are shader-clock-domain ticks (may be a constant 100 MHz timer): compare ratios ==\n");
  // one wave per SIMD (256 threads = 4 waves, 1 block/CU) and two waves per SIMD (512 threads)
  for (int thr : {256, 512}) {
    run_mix<4, 0, false>(d_out, 256, thr, "4 MFMA only");
    run_mix<0, 8, false>(d_out, 256, thr, "8 v_fma only");
    run_mix<0, 8, true>(d_out, 256, thr, "8 v_pk_fma only");
    run_mix<4, 2, false>(d_out, 256, thr, "4 x (MFMA + 2 indep v_fma)");
    run_mix<4, 4, false>(d_out, 256, thr, "4 x (MFMA + 4 indep v_fma)");
    run_mix<4, 8, false>(d_out, 256, thr, "4 x (MFMA + 8 indep v_fma)");
    run_mix<4, 4, true>(d_out, 256, thr, "4 x (MFMA + 4 indep v_pk_fma)");
  }
  for (int thr : {256, 512}) {
    const int iters = 500;
    float h[2];
    hipLaunchKernelGGL((k_dep<0>), dim3(256), dim3(thr), 0, 0, d_out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize(); hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
    printf("16 x (MFMA -> 4 dependent |z| FMA)           thr=%3d  cycles/tile %8.1f\n", thr, h[1] / iters);
    hipLaunchKernelGGL((k_dep<4>), dim3(256), dim3(thr), 0, 0, d_out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize(); hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
    printf("16 x (MFMA -> 4 dep |z| FMA + 4 more v_fma)  thr=%3d  cycles/tile %8.1f\n", thr, h[1] / iters);
  }
  // wall-clock versions (event timed), whole chip
  auto wall = [&](auto kern, int thr, int iters, const char* name, double flop_per_iter_per_wave) {
    float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(256 * (thr == 256 ? 2 : 1)), dim3(thr), 0, 0, d_out, iters, 1.0001f, 0.5f); }, 5);
    const double waves = 256.0 * 2 * (thr == 256 ? 1 : 1) * (thr / 64) / (thr == 256 ? 1 : 2) ;
    printf("%-44s thr=%3d wall %8.1f us  -> %7.1f TFLOP/s\n", name, thr, us, flop_per_iter_per_wave * iters * waves / us * 1e-6);
  };
  wall(k_mix<4, 0, false>, 256, 20000, "wall: 4 MFMA only (8 waves/CU)", 4 * 2048.0);
  wall(k_mix<0, 8, false>, 256, 20000, "wall: 8 v_fma only", 8 * 128.0);
  wall(k_mix<0, 8, true>, 256, 20000, "wall: 8 v_pk_fma only", 8 * 256.0);
  wall(k_mix<4, 4, false>, 256, 20000, "wall: 4 x (MFMA + 4 v_fma)", 4 * 2048.0 + 16 * 128.0);
  wall(k_mix<4, 8, false>, 256, 20000, "wall: 4 x (MFMA + 8 v_fma)", 4 * 2048.0 + 32 * 128.0);
  wall(k_mix<4, 4, true>, 256, 20000, "wall: 4 x (MFMA + 4 v_pk_fma)", 4 * 2048.0 + 16 * 256.0);
  // start-up costs
  printf("empty kernel, 512 blocks x 256: %.2f us per launch (back-to-back)\n",
         time_us([&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, 0, d_out); }, 200));
  printf("130 L2-resident per-lane loads, 512 blocks x 256: %.2f us per launch\n",
         time_us([&] { hipLaunchKernelGGL(k_consts, dim3(512), dim3(256), 0, 0, d_w, d_out, 4096); }, 200));
  printf("130 per-lane loads, 2048 blocks x 256: %.2f us per launch\n",
         time_us([&] { hipLaunchKernelGGL(k_consts, dim3(2048), dim3(256), 0, 0, d_w, d_out, 4096); }, 200));
  return 0;
}
