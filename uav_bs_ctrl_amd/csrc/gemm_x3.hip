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

// BM_ = 128: four waves of 64 x 64 (2 x 2 tiles of 32 x 32).  BM_ = 64 (UAVGNN_GEMM_TILE_64): 64 x 128 output tile, four waves of
// 64 x 32 - twice the workgroups for batches of a few thousand rows ([4096, 768] x [768, 256] is 64 workgroups of 128 x 128 on 256 CUs).
template <bool ACC, bool RELU, int BM_>
__global__ __launch_bounds__(256, 2) void gemm_nt_x3_kernel(const float* __restrict__ X, int ldx, int M, int K,
                                                            const unsigned short* __restrict__ Bp, int N,
                                                            const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                            int row_blocks, int col_blocks) {
  constexpr int NB = BM_ == 128 ? 2 : 1;     // column tiles of a wave
  constexpr int XI = BM_ / 32;               // float4 of X per thread and slice
  __shared__ u32x4 sA[3 * PT], sB[3 * PT];   // [plane][row][4 chunks of 8 bf16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, lh = lane >> 5, sw = swz32(l32);
  const int wm = BM_ == 128 ? (wave >> 1) * 64 : 0, wn = BM_ == 128 ? (wave & 1) * 64 : wave * 32;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / col_blocks) * 8 + xcd, cb = slot - (slot / col_blocks) * col_blocks;
  if (rb >= row_blocks) return;
  const int m0 = rb * BM_, n0 = cb * BN;

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // X loader: float4 q = tid + 256 i -> row tid / 8 + 32 i, k = 4 (tid % 8); rows past M are clamped (stores are masked)
  const int lr = tid >> 3, c4 = tid & 7;
  unsigned xo[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) xo[i] = static_cast<unsigned>(min(m0 + lr + 32 * i, M - 1)) * ldx + 4 * c4;
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
  float4 ra[XI];
  u32x4 rw[6];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XI; ++i) ra[i] = *reinterpret_cast<const float4*>(X + (xo[i] + k0));
#pragma unroll
    for (int i = 0; i < 6; ++i) rw[i] = *reinterpret_cast<const u32x4*>(Bp + (bo[i] + k0));
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XI; ++i) stage4(sa_w + 32 * i * 32, PT * 8, ra[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) sB[sbw[i]] = rw[i];
    __syncthreads();
    gload(min(k0 + BK, K - BK));                       // unconditional (the tail re-reads the last slice): static vmcnt
    __builtin_amdgcn_sched_barrier(0);                 // keep the loads ahead of the MFMA block
    bf16x8 fa[2][2][3], fb[NB][2][3];                  // [tile][half][plane]
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a][kh][pl] = as_frag(sA[pl * PT + (wm + a * 32 + l32) * 4 + ((2 * kh + lh) ^ sw)]);
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[b][kh][pl] = as_frag(sB[pl * PT + (wn + b * 32 + l32) * 4 + ((2 * kh + lh) ^ sw)]);
      }
    // six products, smallest first; four (BM_ = 64: two x two halves) independent accumulators between dependent MFMAs
#define UAVGNN_X3_TERM(ia, ib)                                                                     \
  _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int a = 0; a < 2; ++a)  \
      _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[a][b] = mfma32(fa[a][kh][ia], fb[b][kh][ib], acc[a][b]);
    UAVGNN_X3_TERM(0, 2) UAVGNN_X3_TERM(2, 0) UAVGNN_X3_TERM(1, 1) UAVGNN_X3_TERM(0, 1) UAVGNN_X3_TERM(1, 0) UAVGNN_X3_TERM(0, 0)
#undef UAVGNN_X3_TERM
  }
  // D layout of a 32 x 32 tile: lane l holds column l % 32, register i holds row 8 (i / 4) + 4 (l / 32) + i % 4
#pragma unroll
  for (int b = 0; b < NB; ++b) {
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

// ---- eight wavefronts: 256 x 128 output tile, LDS double-buffered, ONE barrier per slice (the structure of gru_x3.hip) ---------
// Waves of 64 x 64 (2 x 2 tiles of 32 x 32, 64 accumulator registers); slice t + 1 is split and written to the other buffer
// in the same straight-line code as the MFMAs of slice t, the loads of slice t + 2 are in flight, the fragment reads are
// software-pipelined over the two 16-wide halves of a slice; waves w / w + 4 (same SIMD) stage before / after their first
// MFMA group.  148 KB of LDS: one workgroup per CU.
namespace w8 {
constexpr int BM8 = 256, NT = 512;
constexpr int PA = BM8 * 4, PB = BN * 4;               // 16-byte chunks per split plane of the X / B tile
constexpr int BUF = 3 * PA + 3 * PB;                   // chunks per buffer (72 KB)
}  // namespace w8

template <bool ACC, bool RELU, bool IL>
__global__ __launch_bounds__(w8::NT) void gemm_nt_x3w8_kernel(const float* __restrict__ X, int ldx, int M, int K,
                                                              const unsigned short* __restrict__ Bp, int N,
                                                              const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                              int row_blocks, int col_blocks, const float* __restrict__ X2, int ldx2,
                                                              int ns1, float* __restrict__ rowmax_out) {
  // rowmax_out != nullptr: max |.| over every row of X is written there by the workgroups of column block 0 (every element of the
  // row passes through their staging registers): the producer-side bound of the f16x2 weight gradient behind this layer.
  // X2 != nullptr: the contraction runs over [X || X2] (two buffers, no concatenated copy): slices 0 .. ns1 - 1 come from X, the
  // rest from X2 (uavgnn_gemm_nt_x3_cat).  One source: ns1 = K / 32.
  using namespace w8;
  __shared__ u32x4 smem[2 * BUF];   // buffer b: X planes [3][256][4] then B planes [3][128][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, lh = lane >> 5, sw = swz32(l32);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / col_blocks) * 8 + xcd, cb = slot - (slot / col_blocks) * col_blocks;
  if (rb >= row_blocks) return;
  const int m0 = rb * BM8, n0 = cb * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // X loader: float4 q = tid + 512 i (i < 4) -> row tid / 8 + 64 i, k = 4 (tid % 8)
  const int lr = tid >> 3, c4 = tid & 7;
  unsigned xo[4], xo2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xo[i] = static_cast<unsigned>(min(m0 + lr + 64 * i, M - 1)) * ldx + 4 * c4;
    xo2[i] = static_cast<unsigned>(min(m0 + lr + 64 * i, M - 1)) * ldx2 + 4 * c4;
  }
  const int sa_w = lr * 32 + (((c4 >> 1) ^ swz32(lr)) * 8) + (c4 & 1) * 4;   // bf16 units inside an X plane
  // B loader: chunk q = tid + 512 i (i < 3): plane i, row tid / 4, chunk tid % 4
  unsigned bo[3];
  int sbw[3];
  const unsigned plane = static_cast<unsigned>(N) * K;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = tid >> 2, c = tid & 3;
    bo[i] = i * plane + static_cast<unsigned>(min(n0 + row, N - 1)) * K + 8 * c;
    sbw[i] = 3 * PA + i * PB + row * 4 + (c ^ swz32(row));
  }
  const int ns = K / BK;
  float4 ra[4];
  u32x4 rw[3];
  auto gload_a = [&](int t) {
    const bool second = t >= ns1;                                  // wave-uniform
    const float* __restrict__ src = second ? X2 : X;
    const unsigned k0 = static_cast<unsigned>(second ? t - ns1 : t) * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(src + ((second ? xo2[i] : xo[i]) + k0));
  };
  auto gload_w = [&](int t) {
    const unsigned k0 = static_cast<unsigned>(t) * BK;
#pragma unroll
    for (int i = 0; i < 3; ++i) rw[i] = *reinterpret_cast<const u32x4*>(Bp + (bo[i] + k0));
  };
  auto gload = [&](int t) {
    gload_a(t);
    gload_w(t);
  };
  const bool track = rowmax_out != nullptr && cb == 0;      // workgroup-uniform
  float rmx[4] = {0.f, 0.f, 0.f, 0.f};
  auto lstore_a = [&](int buf, int i) {
    unsigned short* sa = reinterpret_cast<unsigned short*>(smem + buf * BUF) + sa_w;
    stage4(sa + 64 * i * 32, PA * 8, ra[i]);
    if (track) rmx[i] = fmaxf(fmaxf(fmaxf(rmx[i], fabsf(ra[i].x)), fmaxf(fabsf(ra[i].y), fabsf(ra[i].z))), fabsf(ra[i].w));
  };
  auto lstore_b = [&](int buf) {
    u32x4* sb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 3; ++i) sb[sbw[i]] = rw[i];
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) lstore_a(buf, i);
    lstore_b(buf);
  };
  struct Half {
    bf16x8 a[2][3], b[2][3];   // [tile][plane]
  };
#define UAVGNN_X3_READ(F, buf, kh)                                                                                 \
  {                                                                                                                \
    const u32x4* sb = smem + (buf) * BUF;                                                                          \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {               \
      F.a[a][pl] = as_frag(sb[pl * PA + (wm + a * 32 + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);                        \
      F.b[a][pl] = as_frag(sb[3 * PA + pl * PB + (wn + a * 32 + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);               \
    }                                                                                                              \
  }
#define UAVGNN_X3_TERM(ia, ib)                                                                      \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)      \
      acc[a][b] = mfma32(F.a[a][ia], F.b[b][ib], acc[a][b]);
#define UAVGNN_X3_MFMA(F_)                                                                           \
  {                                                                                                  \
    const Half& F = F_;                                                                              \
    UAVGNN_X3_TERM(0, 2) UAVGNN_X3_TERM(2, 0) UAVGNN_X3_TERM(1, 1) UAVGNN_X3_TERM(0, 1) UAVGNN_X3_TERM(1, 0) UAVGNN_X3_TERM(0, 0) \
  }
  gload(0);
  lstore(0);
  gload(min(1, ns - 1));
  __syncthreads();
  const bool early = wave < 4;
  Half f0, f1;
  UAVGNN_X3_READ(f0, 0, 0)
  if (IL) {
    // The staging of slice t + 1 (split VALU, LDS stores) and the loads of slice t + 2 are INTERLEAVED with the first MFMA group
    // in program order (sched_group_barrier): an independent VALU / LDS / memory instruction issues in the shadow of an
    // executing MFMA only when it follows it in the instruction stream - as a block in front of the MFMAs its issue time adds to
    // theirs (tools/ubench/mfma_bf16.hip).  The weight-plane loads go out right behind the LDS stores of their registers.
    for (int t = 0; t < ns; ++t) {
      UAVGNN_X3_READ(f1, t & 1, 1)
      __builtin_amdgcn_sched_barrier(0);
      const Half& F = f0;
      const int tn = min(t + 2, ns - 1);
      lstore_b((t + 1) & 1);
      gload_w(tn);
      UAVGNN_X3_TERM(0, 2)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) lstore_a((t + 1) & 1, i);
      UAVGNN_X3_TERM(2, 0) UAVGNN_X3_TERM(1, 1) UAVGNN_X3_TERM(0, 1) UAVGNN_X3_TERM(1, 0)
#pragma unroll
      for (int sg = 0; sg < 16; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      gload_a(tn);
      UAVGNN_X3_TERM(0, 0)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      UAVGNN_X3_READ(f0, (t + 1) & 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_X3_MFMA(f1)
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int t = 0; t < ns; ++t) {
      UAVGNN_X3_READ(f1, t & 1, 1)
      if (early) {
        lstore((t + 1) & 1);                 // slice t + 1 (the tail re-stages the last slice: unconditional, straight-line)
        gload(min(t + 2, ns - 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_X3_MFMA(f0)
      __builtin_amdgcn_sched_barrier(0);
      if (!early) {
        lstore((t + 1) & 1);
        gload(min(t + 2, ns - 1));
      }
      __syncthreads();
      UAVGNN_X3_READ(f0, (t + 1) & 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_X3_MFMA(f1)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef UAVGNN_X3_MFMA
#undef UAVGNN_X3_TERM
#undef UAVGNN_X3_READ
  if (track) {      // the eight threads of a row (consecutive lanes) hold the maxima of their 4-float pieces
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float m = rmx[i];
      m = fmaxf(m, __shfl_xor(m, 1));
      m = fmaxf(m, __shfl_xor(m, 2));
      m = fmaxf(m, __shfl_xor(m, 4));
      if (c4 == 0 && m0 + lr + 64 * i < M) rowmax_out[m0 + lr + 64 * i] = m;
    }
  }
  // (round 6: the NON-accumulating instantiations take the tile path too - 16 row-contiguous float4 stores per lane instead of 64 dword
  // stores straight from the D layout: -3 us per tile, measured on the f16x2 twin of this kernel, csrc/gemm_h2.hip)
  if ((ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && n0 + BN <= N) {
    // Y (+)= product through an LDS tile: the accumulators are parked in LDS (the slice buffers are free behind the loop's last
    // barrier), then every thread does a ROW-CONTIGUOUS float4 read-modify-write of Y - its 16 loads are issued back to back,
    // one exposed round trip per tile.  Straight from the D layout (`v += *p` per element) the 64 dword loads of a lane are
    // issued in register-sized batches, each waiting for its own round trip: the accumulating launch of the GRU backward (dh +=
    // d_gh W_hh) took 103 us against 76 us for the same product without the read.
    constexpr int LDT = BN + 4;
    float* sT = reinterpret_cast<float*>(smem);           // [256][LDT] fp32
    static_assert(BM8 * LDT * 4 <= 2 * BUF * 16, "the output tile fits the slice buffers");
    __syncthreads();                                     // the last fragment reads of the stale buffer
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float bv = bias != nullptr ? bias[n0 + wn + b * 32 + l32] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          sT[(wm + a * 32 + 8 * (i >> 2) + 4 * lh + (i & 3)) * LDT + wn + b * 32 + l32] = acc[a][b][i] + bv;
      }
    float4 yv[16];
    if constexpr (ACC) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + NT * q, row = idx >> 5, c4 = idx & 31;
        yv[q] = *reinterpret_cast<const float4*>(Y + static_cast<size_t>(min(m0 + row, M - 1)) * ldy + n0 + 4 * c4);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = tid + NT * q, row = idx >> 5, c4 = idx & 31;
      const float4 t = *reinterpret_cast<const float4*>(sT + row * LDT + 4 * c4);
      float4 o = t;
      if constexpr (ACC) o = {yv[q].x + t.x, yv[q].y + t.y, yv[q].z + t.z, yv[q].w + t.w};
      if (RELU) o = {fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
      if (m0 + row < M) *reinterpret_cast<float4*>(Y + static_cast<size_t>(m0 + row) * ldy + n0 + 4 * c4) = o;
    }
    return;
  }
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

// Kernel variants are selected PER CALL by bits of the `epilogue` word (no process-wide state: the library is re-entrant):
//   default                          256 x 128 tiles, eight waves, the staging of a slice as a block in front of its MFMAs;
//   UAVGNN_GEMM_STAGING_INTERLEAVED  the same with the staging interleaved with the MFMAs: 7-15 % faster launch by launch, but a
//                                    C3 cycle runs at the package power limit and is 0.5-2 ms SLOWER with it
//                                    (profiles/r03_staging_interleave.txt);
//   UAVGNN_GEMM_TILE_128             128 x 128 tiles, four waves (the round-2 kernel).

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

// K1 = columns taken from X (a multiple of 32), the remaining K - K1 from X2 (nullptr: one source, K1 = K)
static int gemm_nt_x3_launch(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const void* planes, int N,
                             const float* bias, float* Y, int ldy, int epilogue, uavgnn_stream_t stream, float* rowmax_out = nullptr) {
  if (rowmax_out != nullptr && (X2 != nullptr || (epilogue & (UAVGNN_GEMM_TILE_128 | UAVGNN_GEMM_TILE_64)))) return UAVGNN_EUNSUPPORTED;
  if (M < 0 || !X || !planes || !Y || ldx < K1 || ldy < N || K1 <= 0 || K1 > K || (X2 == nullptr) != (K1 == K) ||
      (X2 != nullptr && ldx2 < K - K1))
    return UAVGNN_EINVAL;
  if (M == 0) return 0;
  if (!uavgnn_gemm_x3_supported(M, N, K) || (ldx & 3) || static_cast<long long>(M) * ldx >= (1LL << 31) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(planes)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (X2 != nullptr && ((K1 % BK) || (ldx2 & 3) || static_cast<long long>(M) * ldx2 >= (1LL << 31) ||
                        (reinterpret_cast<uintptr_t>(X2) & 15) || (epilogue & (UAVGNN_GEMM_TILE_128 | UAVGNN_GEMM_TILE_64))))
    return UAVGNN_EUNSUPPORTED;          // two sources: the eight-wave kernel only
  const int ns1 = K1 / BK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short* bp = static_cast<const unsigned short*>(planes);
  const bool acc = (epilogue & UAVGNN_GEMM_ACCUMULATE) != 0, relu = (epilogue & UAVGNN_GEMM_RELU) != 0;
  const int col_blocks = (N + BN - 1) / BN;
  if (!(epilogue & (UAVGNN_GEMM_TILE_128 | UAVGNN_GEMM_TILE_64))) {
    const int row_blocks = (M + w8::BM8 - 1) / w8::BM8;
    const dim3 grid(((row_blocks + 7) / 8) * 8 * col_blocks), block(w8::NT);
#define UAVGNN_X3_GEMM(ACC, RELU, IL)                                                                                        \
  hipLaunchKernelGGL((gemm_nt_x3w8_kernel<ACC, RELU, IL>), grid, block, 0, st, X, ldx, M, K, bp, N, bias, Y, ldy, row_blocks, \
                     col_blocks, X2, ldx2, ns1, rowmax_out)
#define UAVGNN_X3_GEMM_IL(IL)                         \
  if (acc && relu) UAVGNN_X3_GEMM(true, true, IL);    \
  else if (acc) UAVGNN_X3_GEMM(true, false, IL);      \
  else if (relu) UAVGNN_X3_GEMM(false, true, IL);     \
  else UAVGNN_X3_GEMM(false, false, IL);
    if (epilogue & UAVGNN_GEMM_STAGING_INTERLEAVED) { UAVGNN_X3_GEMM_IL(true) }
    else { UAVGNN_X3_GEMM_IL(false) }
#undef UAVGNN_X3_GEMM_IL
#undef UAVGNN_X3_GEMM
    return launch_status();
  }
  const int bm = (epilogue & UAVGNN_GEMM_TILE_64) ? 64 : BM;
  const int row_blocks = (M + bm - 1) / bm;
  const dim3 grid(((row_blocks + 7) / 8) * 8 * col_blocks), block(256);
#define UAVGNN_X3_GEMM(ACC, RELU, BM_)                                                                                       \
  hipLaunchKernelGGL((gemm_nt_x3_kernel<ACC, RELU, BM_>), grid, block, 0, st, X, ldx, M, K, bp, N, bias, Y, ldy, row_blocks, \
                     col_blocks)
#define UAVGNN_X3_GEMM_BM(BM_)                         \
  if (acc && relu) UAVGNN_X3_GEMM(true, true, BM_);    \
  else if (acc) UAVGNN_X3_GEMM(true, false, BM_);      \
  else if (relu) UAVGNN_X3_GEMM(false, true, BM_);     \
  else UAVGNN_X3_GEMM(false, false, BM_);
  if (bm == 64) { UAVGNN_X3_GEMM_BM(64) }
  else { UAVGNN_X3_GEMM_BM(128) }
#undef UAVGNN_X3_GEMM_BM
#undef UAVGNN_X3_GEMM
  return launch_status();
}

extern "C" int uavgnn_gemm_nt_x3(const float* X, int ldx, int M, int K, const void* planes, int N, const float* bias,
                                 float* Y, int ldy, int epilogue, uavgnn_stream_t stream) {
  if (ldx < K) return UAVGNN_EINVAL;
  return gemm_nt_x3_launch(X, ldx, K, nullptr, 0, M, K, planes, N, bias, Y, ldy, epilogue, stream);
}

// uavgnn_gemm_nt_x3 (eight-wave kernel only: no tile variants) that ALSO writes rowmax_out [M] = max |.| over every row of X - every
// element of a row passes through the staging registers of the row's first column block: the bound of an f16x2 product that reads the
// same operand later (the weight gradient of the layer: uavgnn_gemm_tn_h2), at no extra traffic.
extern "C" int uavgnn_gemm_nt_x3_rowmax(const float* X, int ldx, int M, int K, const void* planes, int N, const float* bias, float* Y,
                                        int ldy, int epilogue, float* rowmax_out, uavgnn_stream_t stream) {
  if (ldx < K || !rowmax_out) return UAVGNN_EINVAL;
  return gemm_nt_x3_launch(X, ldx, K, nullptr, 0, M, K, planes, N, bias, Y, ldy, epilogue, stream, rowmax_out);
}

// Y = [X (K1 columns) || X2 (K - K1 columns)] B^T ...: the contraction over two buffers without a concatenated copy (the GRU
// backward's d x = d_gi W_ih[:, :H] + d_proj Wp[:, :H] as ONE product; planes = the split of the stacked weight [K, N]^T)
extern "C" int uavgnn_gemm_nt_x3_cat(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const void* planes,
                                     int N, const float* bias, float* Y, int ldy, int epilogue, uavgnn_stream_t stream) {
  if (!X2) return UAVGNN_EINVAL;
  return gemm_nt_x3_launch(X, ldx, K1, X2, ldx2, M, K, planes, N, bias, Y, ldy, epilogue, stream);
}
