// The dense layers of the path on the bf16 matrix cores (bf16x3.h): Y[M, N] = X[M, K] * B[N, K]^T (+ bias) (+ Y), fp32 in and
// out, every fp32 product accumulated as six exact bf16 x bf16 MFMA products (error below an fp32 GEMM's on the same data,
// profiles/r02_ubench_gemm_bf16x3.txt).
// Replaces the nn.Linear GEMMs of /root/reference/algos/madrqn/agents/gnn_agents.py (f_aggr :101-102, :106; TarMAC f_val /
// f_sign / f_que :227-236; the GRU input / hidden GEMMs of the BPTT backward; f_out :30) and their input-gradient halves
// under loss.backward() (learner.py:157):
//   forward      y  = x W^T + b      B = W  [out, in]   -> uavgnn_split_bf16x3(W, transpose = 0)
//   input grad   dx = dy W           B = W^T [in, out]  -> uavgnn_split_bf16x3(W, transpose = 1)
// (weight gradients dW = dy^T x contract over the agent axis and stay on the vendor's split-K batched GEMM.)
//
// One workgroup: 128 x 128 output tile, four wavefronts of 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16 tiles, 64 accumulator
// registers); per 32-wide K slice the X tile is loaded as fp32, split in registers and written to LDS as three bf16 planes,
// the pre-split B planes are copied; LDS rows are 64 B with the XOR swizzle of bf16x3.h (conflict-free ds_read_b128
// fragments), 48 KB per workgroup, two to three workgroups per CU; the next slice is in flight in registers while the current
// one computes.  The column blocks of a row block run on the same XCD (X is re-read from that XCD's L2).
#include "bf16x3.h"
#include "common.h"

namespace uavgnn {
namespace {

using namespace x3;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PT = 128 * 4;   // 16-byte chunks per split plane of a 128-row tile

// W [R, C] (row stride ld) -> planes [3][R][C], or [3][C][R] when transposed; one thread per output pair
__global__ __launch_bounds__(256) void split_matrix_kernel(const float* __restrict__ W, int ld, int R, int C, int transpose,
                                                           unsigned short* __restrict__ planes) {
  const int orow_n = transpose ? C : R, ocol_n = transpose ? R : C;
  const long long n = static_cast<long long>(orow_n) * ocol_n;
  const long long p = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 2;
  if (p >= n) return;
  const int orow = static_cast<int>(p / ocol_n), ocol = static_cast<int>(p - static_cast<long long>(orow) * ocol_n);
  float x, y;
  if (transpose) {
    x = W[static_cast<size_t>(ocol) * ld + orow];
    y = W[static_cast<size_t>(ocol + 1) * ld + orow];
  } else {
    x = W[static_cast<size_t>(orow) * ld + ocol];
    y = W[static_cast<size_t>(orow) * ld + ocol + 1];
  }
  const Split3 s = split_pair(x, y);
  *reinterpret_cast<unsigned*>(planes + p) = s.h1;
  *reinterpret_cast<unsigned*>(planes + n + p) = s.h2;
  *reinterpret_cast<unsigned*>(planes + 2 * n + p) = s.h3;
}

template <bool ACC, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_nt_x3_kernel(const float* __restrict__ X, int ldx, int M, int K,
                                                            const unsigned short* __restrict__ Bp, int N,
                                                            const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                            int row_blocks, int col_blocks) {
  __shared__ u32x4 sA[3 * PT], sB[3 * PT];   // [plane][row][4 chunks of 8 bf16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, lh = lane >> 5, sw = swz32(l32);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / col_blocks) * 8 + xcd, cb = slot - (slot / col_blocks) * col_blocks;
  if (rb >= row_blocks) return;
  const int m0 = rb * BM, n0 = cb * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // X loader: float4 q = tid + 256 i -> row tid / 8 + 32 i, k = 4 (tid % 8); rows past M are clamped (stores are masked)
  const int lr = tid >> 3, c4 = tid & 7;
  unsigned xo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xo[i] = static_cast<unsigned>(min(m0 + lr + 32 * i, M - 1)) * ldx + 4 * c4;
  unsigned short* sa_w = reinterpret_cast<unsigned short*>(sA) + lr * 32 + (((c4 >> 1) ^ swz32(lr)) * 8) + (c4 & 1) * 4;
  // B loader: chunk q = tid + 256 i (i < 6): plane q / 512, row (q % 512) / 4, chunk q % 4; rows past N are clamped
  unsigned bo[6];
  int sbw[6];
  const unsigned plane = static_cast<unsigned>(N) * K;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = tid + 256 * i, pl = q >> 9, row = (q & 511) >> 2, c = q & 3;
    bo[i] = pl * plane + static_cast<unsigned>(min(n0 + row, N - 1)) * K + 8 * c;
    sbw[i] = pl * PT + row * 4 + (c ^ swz32(row));
  }
  float4 ra[4];
  u32x4 rw[6];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(X + (xo[i] + k0));
#pragma unroll
    for (int i = 0; i < 6; ++i) rw[i] = *reinterpret_cast<const u32x4*>(Bp + (bo[i] + k0));
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) stage4(sa_w + 32 * i * 32, PT * 8, ra[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) sB[sbw[i]] = rw[i];
    __syncthreads();
    gload(min(k0 + BK, K - BK));                       // unconditional (the tail re-reads the last slice): static vmcnt
    __builtin_amdgcn_sched_barrier(0);                 // keep the loads ahead of the MFMA block
    bf16x8 fa[2][2][3], fb[2][2][3];                   // [tile][half][plane]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          fa[a][kh][pl] = as_frag(sA[pl * PT + (wm + a * 32 + l32) * 4 + ((2 * kh + lh) ^ sw)]);
          fb[a][kh][pl] = as_frag(sB[pl * PT + (wn + a * 32 + l32) * 4 + ((2 * kh + lh) ^ sw)]);
        }
    // six products, smallest first; four independent accumulators between dependent MFMAs
#define UAVGNN_X3_TERM(ia, ib)                                                                     \
  _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int a = 0; a < 2; ++a)  \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(fa[a][kh][ia], fb[b][kh][ib], acc[a][b]);
    UAVGNN_X3_TERM(0, 2) UAVGNN_X3_TERM(2, 0) UAVGNN_X3_TERM(1, 1) UAVGNN_X3_TERM(0, 1) UAVGNN_X3_TERM(1, 0) UAVGNN_X3_TERM(0, 0)
#undef UAVGNN_X3_TERM
  }
  // D layout of a 32 x 32 tile: lane l holds column l % 32, register i holds row 8 (i / 4) + 4 (l / 32) + i % 4
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + wn + b * 32 + l32;
    if (col >= N) continue;
    const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm + a * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
        if (row < M) {
          float* p = Y + static_cast<size_t>(row) * ldy + col;
          float v = acc[a][b][i] + bv;
          if (ACC) v += *p;
          if (RELU) v = fmaxf(v, 0.f);
          *p = v;
        }
      }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_gemm_x3_supported(int M, int N, int K) {
  // 32-bit element offsets inside the kernel: every operand below 2^31 elements
  return (M > 0 && N > 0 && K >= BK && K % BK == 0 && static_cast<long long>(M) * K < (1LL << 31) &&
          3LL * N * K < (1LL << 31)) ? 1 : 0;
}

extern "C" int uavgnn_split_bf16x3(const float* W, int ld, int R, int C, int transpose, void* planes,
                                   uavgnn_stream_t stream) {
  if (!W || !planes || R <= 0 || C <= 0 || ld < C) return UAVGNN_EINVAL;
  const int ocol_n = transpose ? R : C;
  if ((ocol_n & 1) || (reinterpret_cast<uintptr_t>(planes) & 15)) return UAVGNN_EUNSUPPORTED;
  const long long pairs = static_cast<long long>(R) * C / 2;
  hipLaunchKernelGGL(split_matrix_kernel, dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W, ld, R, C, transpose, static_cast<unsigned short*>(planes));
  return launch_status();
}

extern "C" int uavgnn_gemm_nt_x3(const float* X, int ldx, int M, int K, const void* planes, int N, const float* bias,
                                 float* Y, int ldy, int epilogue, uavgnn_stream_t stream) {
  if (M < 0 || !X || !planes || !Y || ldx < K || ldy < N) return UAVGNN_EINVAL;
  if (M == 0) return 0;
  if (!uavgnn_gemm_x3_supported(M, N, K) || (ldx & 3) || static_cast<long long>(M) * ldx >= (1LL << 31) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(planes)) & 15))
    return UAVGNN_EUNSUPPORTED;
  const int row_blocks = (M + BM - 1) / BM, col_blocks = (N + BN - 1) / BN;
  const dim3 grid(((row_blocks + 7) / 8) * 8 * col_blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short* bp = static_cast<const unsigned short*>(planes);
#define UAVGNN_X3_GEMM(ACC, RELU)                                                                                      \
  hipLaunchKernelGGL((gemm_nt_x3_kernel<ACC, RELU>), grid, block, 0, st, X, ldx, M, K, bp, N, bias, Y, ldy, row_blocks, \
                     col_blocks)
  const bool acc = (epilogue & UAVGNN_GEMM_ACCUMULATE) != 0, relu = (epilogue & UAVGNN_GEMM_RELU) != 0;
  if (acc && relu) UAVGNN_X3_GEMM(true, true);
  else if (acc) UAVGNN_X3_GEMM(true, false);
  else if (relu) UAVGNN_X3_GEMM(false, true);
  else UAVGNN_X3_GEMM(false, false);
#undef UAVGNN_X3_GEMM
  return launch_status();
}
