// A/B for VERDICT r1 item 4 (fused GRU cell): can a hand-written fp32-MFMA GEMM reach the vendor's rate at the GRU shapes?
//   C[M, N] = A[M, K] * W[N, K]^T   (row-major, the nn.Linear / nn.GRUCell layout),  M = 32768 agents,
//   (N, K) = (768, 320) for W_ih [x || c] and (768, 256) for W_hh.
// A fused GRU kernel would have to run its two GEMMs at >= vendor_time / (vendor_time + gate_kernel_time) ~ 0.86 of the
// vendor's rate just to break even; this file measures what a straightforward LDS-tiled v_mfma_f32_16x16x4_f32 kernel
// (128 x 128 x 16 block tile, 64 x 64 per wave, register-prefetched global loads, conflict-free LDS strides) achieves
// next to rocBLAS on the same buffers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_f32.hip -lrocblas -o tools/ubench/bin/gemm_f32
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, BK = 16, LDS_STRIDE = 18;   // 18: banks (18 j + g) mod 32 distinct over a 32-lane group

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         float* __restrict__ C, int M, int N, int K) {
  __shared__ float sA[BM * LDS_STRIDE], sB[BN * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader: thread -> (row = tid / 2, 8 consecutive k) of both tiles
  const int lr = tid >> 1, lk = (tid & 1) * 8;
  const float* ga = A + static_cast<size_t>(m0 + lr) * K + lk;
  const float* gb = W + static_cast<size_t>(n0 + lr) * K + lk;
  float4 ra0 = *reinterpret_cast<const float4*>(ga), ra1 = *reinterpret_cast<const float4*>(ga + 4);
  float4 rb0 = *reinterpret_cast<const float4*>(gb), rb1 = *reinterpret_cast<const float4*>(gb + 4);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    {
      float* pa = sA + lr * LDS_STRIDE + lk;
      float* pb = sB + lr * LDS_STRIDE + lk;
      *reinterpret_cast<float2*>(pa) = make_float2(ra0.x, ra0.y); *reinterpret_cast<float2*>(pa + 2) = make_float2(ra0.z, ra0.w);
      *reinterpret_cast<float2*>(pa + 4) = make_float2(ra1.x, ra1.y); *reinterpret_cast<float2*>(pa + 6) = make_float2(ra1.z, ra1.w);
      *reinterpret_cast<float2*>(pb) = make_float2(rb0.x, rb0.y); *reinterpret_cast<float2*>(pb + 2) = make_float2(rb0.z, rb0.w);
      *reinterpret_cast<float2*>(pb + 4) = make_float2(rb1.x, rb1.y); *reinterpret_cast<float2*>(pb + 6) = make_float2(rb1.z, rb1.w);
    }
    __syncthreads();
    if (k0 + BK < K) {   // next tile in flight while this one computes
      ra0 = *reinterpret_cast<const float4*>(ga + k0 + BK); ra1 = *reinterpret_cast<const float4*>(ga + k0 + BK + 4);
      rb0 = *reinterpret_cast<const float4*>(gb + k0 + BK); rb1 = *reinterpret_cast<const float4*>(gb + k0 + BK + 4);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      float fa[4], fb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = sA[(wm + a * 16 + j) * LDS_STRIDE + ks + g];
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = sB[(wn + b * 16 + j) * LDS_STRIDE + ks + g];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }
  // D layout: lane (g, j) holds rows 4g..4g+3 (A-tile rows), column j (B-tile row = output column)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[static_cast<size_t>(m0 + wm + a * 16 + 4 * g + r) * N + n0 + wn + b * 16 + j] = acc[a][b][r];
}


// Variant 2: lane group g owns k in {4g..4g+3} of the BK = 16 slice (k-step s uses k = 4g + s for BOTH operands - the
// contraction order is free), so one ds_read_b128 fetches a fragment's values for all four k-steps: 8 LDS reads per 64
// MFMAs instead of 32.  LDS rows padded to 20 floats (16-byte aligned rows).
constexpr int S2 = 20;
template <int BKK>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel2(const float* __restrict__ A, const float* __restrict__ W,
                                                          float* __restrict__ C, int M, int N, int K) {
  constexpr int ST = BKK + 4;
  __shared__ __attribute__((aligned(16))) float sA[BM * ST], sB[BN * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int F4 = BKK / 4;                 // float4 per tile row
  constexpr int PER = BM * F4 / 256;          // float4 per thread per operand
  float4 ra[PER], rb[PER];
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = tid + 256 * q, row = idx / F4, c4 = idx % F4;
      ra[q] = *reinterpret_cast<const float4*>(A + static_cast<size_t>(m0 + row) * K + k0 + 4 * c4);
      rb[q] = *reinterpret_cast<const float4*>(W + static_cast<size_t>(n0 + row) * K + k0 + 4 * c4);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BKK) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = tid + 256 * q, row = idx / F4, c4 = idx % F4;
      *reinterpret_cast<float4*>(sA + row * ST + 4 * c4) = ra[q];
      *reinterpret_cast<float4*>(sB + row * ST + 4 * c4) = rb[q];
    }
    __syncthreads();
    if (k0 + BKK < K) gload(k0 + BKK);
#pragma unroll
    for (int kk = 0; kk < BKK; kk += 16) {
      f32x4 fa[4], fb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const f32x4*>(sA + (wm + a * 16 + j) * ST + kk + 4 * g);
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = *reinterpret_cast<const f32x4*>(sB + (wn + b * 16 + j) * ST + kk + 4 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[static_cast<size_t>(m0 + wm + a * 16 + 4 * g + r) * N + n0 + wn + b * 16 + j] = acc[a][b][r];
}

// Variant 3: variant 1 with a BK = 32 slice (half the barriers) and conflict-free stride 34; optional 32x32x2 MFMA.
template <bool MF32>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel3(const float* __restrict__ A, const float* __restrict__ W,
                                                          float* __restrict__ C, int M, int N, int K) {
  constexpr int BKK = 32, ST = 34;
  __shared__ __attribute__((aligned(16))) float sA[BM * ST], sB[BN * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // loader: thread -> (row = tid / 2, 16 consecutive k)
  const int lr = tid >> 1, lk = (tid & 1) * 16;
  const float* ga = A + static_cast<size_t>(m0 + lr) * K + lk;
  const float* gb = W + static_cast<size_t>(n0 + lr) * K + lk;
  float4 ra[4], rb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { ra[q] = reinterpret_cast<const float4*>(ga)[q]; rb[q] = reinterpret_cast<const float4*>(gb)[q]; }
  if constexpr (!MF32) {
    const int j = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += BKK) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* pa = sA + lr * ST + lk + 4 * q;
        float* pb = sB + lr * ST + lk + 4 * q;
        *reinterpret_cast<float2*>(pa) = make_float2(ra[q].x, ra[q].y); *reinterpret_cast<float2*>(pa + 2) = make_float2(ra[q].z, ra[q].w);
        *reinterpret_cast<float2*>(pb) = make_float2(rb[q].x, rb[q].y); *reinterpret_cast<float2*>(pb + 2) = make_float2(rb[q].z, rb[q].w);
      }
      __syncthreads();
      if (k0 + BKK < K) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ra[q] = reinterpret_cast<const float4*>(ga + k0 + BKK)[q];
          rb[q] = reinterpret_cast<const float4*>(gb + k0 + BKK)[q];
        }
      }
#pragma unroll
      for (int ks = 0; ks < BKK; ks += 4) {
        float fa[4], fb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = sA[(wm + a * 16 + j) * ST + ks + g];
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = sB[(wn + b * 16 + j) * ST + ks + g];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[static_cast<size_t>(m0 + wm + a * 16 + 4 * g + r) * N + n0 + wn + b * 16 + j] = acc[a][b][r];
  } else {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int j = lane & 31, g = lane >> 5;      // 32x32x2: operand row / column j, k index g
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += BKK) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* pa = sA + lr * ST + lk + 4 * q;
        float* pb = sB + lr * ST + lk + 4 * q;
        *reinterpret_cast<float2*>(pa) = make_float2(ra[q].x, ra[q].y); *reinterpret_cast<float2*>(pa + 2) = make_float2(ra[q].z, ra[q].w);
        *reinterpret_cast<float2*>(pb) = make_float2(rb[q].x, rb[q].y); *reinterpret_cast<float2*>(pb + 2) = make_float2(rb[q].z, rb[q].w);
      }
      __syncthreads();
      if (k0 + BKK < K) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ra[q] = reinterpret_cast<const float4*>(ga + k0 + BKK)[q];
          rb[q] = reinterpret_cast<const float4*>(gb + k0 + BKK)[q];
        }
      }
#pragma unroll
      for (int ks = 0; ks < BKK; ks += 2) {
        float fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = sA[(wm + a * 32 + j) * ST + ks + g];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = sB[(wn + b * 32 + j) * ST + ks + g];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
    // D layout 32x32: lane (g, j): column j, rows 8*q + 4*g + r for acc[4*q + r]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            C[static_cast<size_t>(m0 + wm + a * 32 + 8 * q + 4 * g + r) * N + n0 + wn + b * 32 + j] = acc[a][b][4 * q + r];
  }
}

int main() {
  const int M = 32768;
  rocblas_handle h;
  rocblas_create_handle(&h);
  for (auto [N, K] : {std::pair<int, int>{768, 320}, {768, 256}, {256, 512}}) {
    float *A, *W, *C, *C2;
    hipMalloc(&A, sizeof(float) * M * K); hipMalloc(&W, sizeof(float) * N * K);
    hipMalloc(&C, sizeof(float) * M * N); hipMalloc(&C2, sizeof(float) * M * N);
    std::vector<float> ha(size_t(M) * K), hw(size_t(N) * K);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = 0.001f * float((i * 7919) % 2003) - 1.f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.002f * float((i * 104729) % 1009) - 1.f;
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    auto mine = [&] { hipLaunchKernelGGL(gemm_nt_kernel, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    auto mine2 = [&] { hipLaunchKernelGGL(gemm_nt_kernel2<16>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    auto mine3 = [&] { hipLaunchKernelGGL(gemm_nt_kernel2<32>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    auto mine4 = [&] { hipLaunchKernelGGL(gemm_nt_kernel3<false>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    auto mine5 = [&] { hipLaunchKernelGGL(gemm_nt_kernel3<true>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    const float one = 1.f, zero = 0.f;
    // row-major C[M,N] = A W^T  ==  column-major C^T[N,M] = W(op T: [N,K]) A^T([K,M]):  sgemm(T, N, N, M, K, W ld K, A ld K, C ld N)
    auto vendor = [&] { rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, W, K, A, K, &zero, C2, N); };
    mine(); vendor();
    hipDeviceSynchronize();
    std::vector<float> c1(size_t(M) * N), c2(size_t(M) * N);
    hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
    double err = 0, mx = 0;
    for (size_t i = 0; i < c1.size(); i += 97) { err = fmax(err, fabs(double(c1[i]) - c2[i])); mx = fmax(mx, fabs(double(c2[i]))); }
    auto time_us = [&](auto f) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) f();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) f();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      return ms * 1000.f / 20;
    };
    const double fl = 2.0 * M * N * K;
    const float t1 = time_us(mine), t2 = time_us(vendor);
    for (int v = 0; v < 2; ++v) {
      if (v == 0) mine4(); else mine5();
      hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
      double e2 = 0;
      for (size_t i = 0; i < c1.size(); i += 97) e2 = fmax(e2, fabs(double(c1[i]) - c2[i]));
      const float tv = v == 0 ? time_us(mine4) : time_us(mine5);
      printf("   variant 3 (BK=32, stride 34, %s): %7.1f us = %6.1f TFLOP/s  (vs rocBLAS %.2f)  max|diff| %.2e\n",
             v == 0 ? "16x16x4 MFMA" : "32x32x2 MFMA", tv, fl / tv * 1e-6, t2 / tv, e2);
    }
    for (int v = 0; v < 2; ++v) {
      if (v == 0) mine2(); else mine3();
      hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
      double e2 = 0;
      for (size_t i = 0; i < c1.size(); i += 97) e2 = fmax(e2, fabs(double(c1[i]) - c2[i]));
      const float tv = v == 0 ? time_us(mine2) : time_us(mine3);
      printf("   variant 2 (k-permuted b128 fragments, BK=%d): %7.1f us = %6.1f TFLOP/s  (vs rocBLAS %.2f)  max|diff| %.2e\n", v == 0 ? 16 : 32, tv,
             fl / tv * 1e-6, t2 / tv, e2);
    }
    printf("M=%d N=%d K=%d  hand-written MFMA GEMM %7.1f us = %6.1f TFLOP/s | rocBLAS %7.1f us = %6.1f TFLOP/s | ratio %.2f | max|diff| %.2e (max|C| %.1f)\n",
           M, N, K, t1, fl / t1 * 1e-6, t2, fl / t2 * 1e-6, t2 / t1, err, mx);
    hipFree(A); hipFree(W); hipFree(C); hipFree(C2);
  }
  return 0;
}
