// fp32 GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 terms (a = a1 + a2 + a3, 8 + 8 + 8
// significand bits) and the product a*b is the sum of the six bf16 x bf16 products whose weight is >= 2^-16 of a1*b1
// (a1b1, a1b2, a2b1, a2b2, a1b3, a3b1); each of those is exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16, the three
// dropped terms are <= 2^-23 |a b| together - the size of ONE fp32 rounding.  fp32 MFMA on gfx950 issues at the fp32 vector
// rate (1/16 of the bf16 MFMA rate, MI355X_MICROARCH.md), so six bf16 MFMAs cost 6/16 of the fp32 MFMA they replace.
//   C[M, N] = A[M, K] * W[N, K]^T   (row-major, nn.Linear layout), M = 32768 agents.
// This file measures the rate AND the error against an fp64 host reference next to rocBLAS sgemm on the same buffers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_bf16x3.hip -lrocblas -o tools/ubench/bin/gemm_bf16x3
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RS = 40;   // LDS row stride in bf16 (80 B): the 16 rows of a b128 fragment read land in 16 distinct 4-bank groups

struct Split3 { unsigned h1, h2, h3; };   // two packed bf16 per word

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

// stage one float4 (4 consecutive k of one row) as three 8-byte bf16 groups
__device__ __forceinline__ void stage4(unsigned short* base, int plane_stride, int off, float4 v) {
  const Split3 a = split_pair(v.x, v.y), b = split_pair(v.z, v.w);
  *reinterpret_cast<uint2*>(base + off) = make_uint2(a.h1, b.h1);
  *reinterpret_cast<uint2*>(base + plane_stride + off) = make_uint2(a.h2, b.h2);
  *reinterpret_cast<uint2*>(base + 2 * plane_stride + off) = make_uint2(a.h3, b.h3);
}

__device__ __forceinline__ bf16x8 frag(const unsigned short* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}

template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x3_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int PL = BM * RS;   // one split plane of a tile (BM == BN)
  __shared__ __attribute__((aligned(16))) unsigned short sA[3 * PL], sB[3 * PL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader: float4 index q = tid + 256 i -> row q / 8, k = 4 (q % 8)
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  const float* ga = A + static_cast<size_t>(m0 + lr) * K + lk;
  const float* gb = W + static_cast<size_t>(n0 + lr) * K + lk;
  float4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const float4*>(ga + static_cast<size_t>(32 * i) * K);
    rb[i] = *reinterpret_cast<const float4*>(gb + static_cast<size_t>(32 * i) * K);
  }
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      stage4(sA, PL, (lr + 32 * i) * RS + lk, ra[i]);
      stage4(sB, PL, (lr + 32 * i) * RS + lk, rb[i]);
    }
    __syncthreads();
    if (k0 + BK < K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(ga + static_cast<size_t>(32 * i) * K + k0 + BK);
        rb[i] = *reinterpret_cast<const float4*>(gb + static_cast<size_t>(32 * i) * K + k0 + BK);
      }
    }
    bf16x8 fb[4][3];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int s = 0; s < 3; ++s) fb[b][s] = frag(sB + s * PL + (wn + b * 16 + j) * RS + 8 * g);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bf16x8 fa[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) fa[s] = frag(sA + s * PL + (wm + a * 16 + j) * RS + 8 * g);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 c = acc[a][b];
        if (TERMS >= 9) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[b][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[b][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[b][1], c, 0, 0, 0);
        }
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[b][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[b][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[b][1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[b][0], c, 0, 0, 0);
        acc[a][b] = c;
      }
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

// ---- variant 2: weights pre-split into bf16 planes [3][N][K] in global memory, XOR-swizzled LDS rows of 64 B (no padding;
// conflict-free for the 16-lane groups ds_read_b128 is served in), loads two K-slices ahead, XCD-aware block order ----------
__global__ void split_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, long long n) {
  const long long i = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  const Split3 s = split_pair(w[i], w[i + 1]);
  *reinterpret_cast<unsigned*>(planes + i) = s.h1;
  *reinterpret_cast<unsigned*>(planes + n + i) = s.h2;
  *reinterpret_cast<unsigned*>(planes + 2 * n + i) = s.h3;
}

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }

template <int TERMS, int PD, int MODE = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16x3_v2(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                            float* __restrict__ C, int M, int N, int K) {
  __shared__ u32x4 sA[3 * 512], sB[3 * 512];   // [plane][row][4 chunks of 8 bf16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int CB = N / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / CB) * 8 + xcd, cb = slot % CB;
  if (rb * BM >= M) return;
  const int m0 = rb * BM, n0 = cb * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lr = tid >> 3, c4 = tid & 7;
  const float* ga = A + static_cast<size_t>(m0 + lr) * K + 4 * c4;
  unsigned short* sa_w = reinterpret_cast<unsigned short*>(sA) + lr * 32 + (((c4 >> 1) ^ swz(lr)) * 8) + (c4 & 1) * 4;
  const size_t plane = static_cast<size_t>(N) * K;
  // B: u32x4 q = tid + 256 i (i < 6): plane q >> 9, row (q & 511) >> 2, chunk q & 3
  const unsigned short* gb[6];
  int sb_w[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = tid + 256 * i, pl = q >> 9, row = (q & 511) >> 2, c = q & 3;
    gb[i] = Wp + pl * plane + static_cast<size_t>(n0 + row) * K + 8 * c;
    sb_w[i] = pl * 512 + row * 4 + (c ^ swz(row));
  }
  float4 ra0[4], ra1[4];
  u32x4 rb0[6], rb1[6];
  auto load = [&](float4 (&ra)[4], u32x4 (&rb4)[6], int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(ga + static_cast<size_t>(32 * i) * K + k0);
#pragma unroll
    for (int i = 0; i < 6; ++i) rb4[i] = *reinterpret_cast<const u32x4*>(gb[i] + k0);
  };
  auto stage = [&](float4 (&ra)[4], u32x4 (&rb4)[6]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stage4(sa_w + 32 * i * 32, 512 * 8, 0, ra[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) sB[sb_w[i]] = rb4[i];
  };
  auto compute = [&]() {
    bf16x8 fb[4][3];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int s = 0; s < 3; ++s) fb[b][s] = __builtin_bit_cast(bf16x8, sB[s * 512 + (wn + b * 16 + j) * 4 + (g ^ swz(j))]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bf16x8 fa[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) fa[s] = __builtin_bit_cast(bf16x8, sA[s * 512 + (wm + a * 16 + j) * 4 + (g ^ swz(j))]);
      // smallest terms first; four independent accumulators between dependent MFMAs
#define UAV_TERM(ia, ib)                                                                                              \
  _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] =                                                           \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ia], fb[b][ib], acc[a][b], 0, 0, 0);
      if (TERMS >= 9) { UAV_TERM(2, 2) UAV_TERM(1, 2) UAV_TERM(2, 1) }
      if (TERMS >= 6) { UAV_TERM(0, 2) UAV_TERM(2, 0) UAV_TERM(1, 1) }
      UAV_TERM(0, 1) UAV_TERM(1, 0) UAV_TERM(0, 0)
#undef UAV_TERM
    }
  };
  const int nk = K / BK;
  // every load is unconditional (the slice index is clamped: the tail re-reads the last slice) so that the number of loads in
  // flight is static and s_waitcnt vmcnt can leave the NEWER set in flight
  const int last = (nk - 1) * BK;
  load(ra0, rb0, 0);
  if (PD == 2) load(ra1, rb1, min(BK, last));
  for (int s = 0; s < nk; s += PD) {
    __syncthreads();
    stage(ra0, rb0);
    __syncthreads();
    if (MODE != 1) load(ra0, rb0, min((s + PD) * BK, last));
    if (MODE != 2) compute();
    if (PD == 2) {   // nk even
      __syncthreads();
      stage(ra1, rb1);
      __syncthreads();
      load(ra1, rb1, min((s + 3) * BK, last));
      compute();
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

int main() {
  const int M = 32768;
  rocblas_handle h;
  rocblas_create_handle(&h);
  for (auto [N, K] : {std::pair<int, int>{768, 320}, {768, 256}, {256, 512}}) {
    float *A, *W, *C, *C2;
    hipMalloc(&A, sizeof(float) * M * K); hipMalloc(&W, sizeof(float) * N * K);
    hipMalloc(&C, sizeof(float) * M * N); hipMalloc(&C2, sizeof(float) * M * N);
    std::vector<float> ha(size_t(M) * K), hw(size_t(N) * K);
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : ha) v = nd(rng);
    for (auto& v : hw) v = nd(rng) * 0.0625f;
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const float one = 1.f, zero = 0.f;
    auto vendor = [&] { rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, W, K, A, K, &zero, C2, N); };
    auto k3 = [&] { hipLaunchKernelGGL(gemm_nt_bf16x3_kernel<3>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    auto k6 = [&] { hipLaunchKernelGGL(gemm_nt_bf16x3_kernel<6>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    auto k9 = [&] { hipLaunchKernelGGL(gemm_nt_bf16x3_kernel<9>, dim3(N / BN, M / BM), dim3(256), 0, 0, A, W, C, M, N, K); };
    unsigned short* Wp;
    hipMalloc(&Wp, sizeof(unsigned short) * 3 * N * K);
    const long long nw = static_cast<long long>(N) * K;
    auto presplit = [&] { hipLaunchKernelGGL(split_bf16x3_kernel, dim3((nw / 2 + 255) / 256), dim3(256), 0, 0, W, Wp, nw); };
    const int grid2 = ((M / BM + 7) / 8) * 8 * (N / BN);
    auto v6p1 = [&] { presplit(); hipLaunchKernelGGL((gemm_nt_bf16x3_v2<6, 1>), dim3(grid2), dim3(256), 0, 0, A, Wp, C, M, N, K); };
    auto v6p2 = [&] { presplit(); hipLaunchKernelGGL((gemm_nt_bf16x3_v2<6, 2>), dim3(grid2), dim3(256), 0, 0, A, Wp, C, M, N, K); };
    auto m1 = [&] { hipLaunchKernelGGL((gemm_nt_bf16x3_v2<6, 1, 1>), dim3(grid2), dim3(256), 0, 0, A, Wp, C, M, N, K); };
    auto m2 = [&] { hipLaunchKernelGGL((gemm_nt_bf16x3_v2<6, 1, 2>), dim3(grid2), dim3(256), 0, 0, A, Wp, C, M, N, K); };
    auto m0 = [&] { hipLaunchKernelGGL((gemm_nt_bf16x3_v2<6, 1, 0>), dim3(grid2), dim3(256), 0, 0, A, Wp, C, M, N, K); };
    auto v3p2 = [&] { presplit(); hipLaunchKernelGGL((gemm_nt_bf16x3_v2<3, 2>), dim3(grid2), dim3(256), 0, 0, A, Wp, C, M, N, K); };
    auto time_us = [&](auto f) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) f();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) f();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      return ms * 1000.f / 20;
    };
    // fp64 reference on a sample of entries
    std::vector<size_t> samp;
    for (size_t i = 0; i < size_t(M) * N; i += 4099) samp.push_back(i);
    std::vector<double> ref(samp.size()), scale(samp.size());
    for (size_t s = 0; s < samp.size(); ++s) {
      const size_t m = samp[s] / N, n = samp[s] % N;
      double acc = 0, sc = 0;
      for (int k = 0; k < K; ++k) { const double p = double(ha[m * K + k]) * double(hw[n * K + k]); acc += p; sc += fabs(p); }
      ref[s] = acc; scale[s] = sc;
    }
    std::vector<float> c(size_t(M) * N);
    auto report = [&](const char* name, float* dev, float t) {
      hipDeviceSynchronize();
      hipMemcpy(c.data(), dev, c.size() * 4, hipMemcpyDeviceToHost);
      double emax = 0, esum = 0;   // error relative to sum_k |a_k b_k| (the quantity fp32 rounding analysis bounds)
      for (size_t s = 0; s < samp.size(); ++s) {
        const double e = fabs(double(c[samp[s]]) - ref[s]) / scale[s];
        emax = fmax(emax, e); esum += e;
      }
      printf("  %-34s %7.1f us = %6.1f TFLOP/s   error / sum|a b|: max %.2e  mean %.2e\n", name, t, 2.0 * M * N * K / t * 1e-6, emax,
             esum / samp.size());
    };
    printf("M=%d N=%d K=%d   (fp32 unit roundoff 2^-24 = 5.96e-08)\n", M, N, K);
    vendor(); report("rocBLAS sgemm (fp32 MFMA)", C2, time_us(vendor));
    k3(); report("bf16 split, 3 products", C, time_us(k3));
    k6(); report("bf16 split, 6 products", C, time_us(k6));
    k9(); report("bf16 split, 9 products", C, time_us(k9));
    v6p1(); report("v2 (pre-split W, swizzle) 6, PD=1", C, time_us(v6p1));
    v6p2(); report("v2 (pre-split W, swizzle) 6, PD=2", C, time_us(v6p2));
    v3p2(); report("v2 (pre-split W, swizzle) 3, PD=2", C, time_us(v3p2));
    printf("  v2 6 PD=1 without the pre-split launch: %.1f us | no global loads in the loop: %.1f us | no LDS reads / MFMA: %.1f us | pre-split launch alone %.1f us\n",
           time_us(m0), time_us(m1), time_us(m2), time_us(presplit));
    hipFree(Wp);
    hipFree(A); hipFree(W); hipFree(C); hipFree(C2);
  }
  return 0;
}
