// Dense layers on the f16 matrix cores with exactly scaled two-term splits ("f16x2", see csrc/gru_h2.hip): Y[M, N] = X[M, K] B[N, K]^T
// (+ bias) (+ Y) (ReLU), fp32 in / out / accumulate, THREE f16 x f16 MFMA products per fp32 product instead of bf16x3's six.
// Replaces the same products as csrc/gemm_x3.hip - the input-gradient halves of the recurrent step under loss.backward()
// (/root/reference/algos/madrqn/learner.py:157 through gnn_agents.py:246, :99): d x = [d_gi || d_proj] [W_ih[:, :H]; Wp[:, :H]],
// d h += d_gh W_hh, and d(K1 output) = d y W_aggr of the time-batched encoder - wherever the kernel that PRODUCED the activation
// operand also hands over its row maxima (the power-of-two row scales; a maximum is order-independent: deterministic):
//   * uavgnn_gru_gates_bwd_fused_sums_rowmax   max over the rows of d_gi and d_gh (one wavefront owns a row of H = 256),
//   * uavgnn_relu_bwd_colsum_rowmax            max over the rows of the masked gradient (one wavefront owns a row of C = 256),
//   * uavgnn_row_absmax                        anything else, as a pass of its own (d_proj: 96 columns).
// Up to two bounds per row (two producers of a two-source operand); the row's scale comes from their maximum.
//
// Arithmetic, error and non-finite behaviour: csrc/gru_h2.hip's header.  B (the weight, [N, K] or its transpose) is split per OUTPUT
// row by uavgnn_split_h2: f16 planes [2][N][K] + 2^-e per row.
//
// Kernel: gemm_x3.hip's eight-wave kernel (256 x 128 output tile, waves of 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16_f16, LDS
// double-buffered with ONE barrier per 32-wide K slice, two-source X loader) with two planes: 48 KB per LDS stage instead of 72, 24
// MFMAs per slice and wavefront instead of 48.
#include "common.h"

namespace uavgnn {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BN = 128, BK = 32, BM8 = 256, NT = 512;
constexpr int PA = BM8 * 4, PB = BN * 4;               // 16-byte chunks per split plane of the X / B tile
constexpr int BUF = 2 * PA + 2 * PB;                   // chunks per buffer (48 KB)

__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ f16x8 as_frag(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int scale_exp(float amax) {   // 2^se * amax in [2^14, 2^15); clamped to the normal range
  const int e = static_cast<int>((__float_as_uint(amax) >> 23) & 0xffu);
  return max(-126, min(126, 14 - (e - 127)));
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float(static_cast<unsigned>(e + 127) << 23); }

struct Split2 {
  unsigned hi, lo;
};
__device__ __forceinline__ Split2 split_pair(float x, float y) {
  Split2 s;
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  s.hi = __builtin_bit_cast(unsigned, h);
  const f32x2 r = f32x2{x, y} - __builtin_convertvector(h, f32x2);
  s.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  return s;
}
__device__ __forceinline__ void stage4(unsigned short* p, int plane_stride, float4 v, float s) {
  const Split2 a = split_pair(v.x * s, v.y * s), b = split_pair(v.z * s, v.w * s);
  *reinterpret_cast<u32x2*>(p) = u32x2{a.hi, b.hi};
  *reinterpret_cast<u32x2*>(p + plane_stride) = u32x2{a.lo, b.lo};
}

// W [R, C] (row stride ld) -> planes [2][R][C] (transpose: [2][C][R]) + winv[rows of the output] = 2^-e: one workgroup per output row
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ W, int ld, int R, int C, int transpose,
                                                       unsigned short* __restrict__ planes, float* __restrict__ winv) {
  __shared__ float red[4];
  const int orow = blockIdx.x, tid = threadIdx.x;
  const int ocol_n = transpose ? R : C;
  const size_t n = static_cast<size_t>(transpose ? C : R) * ocol_n;
  auto at = [&](int j) { return transpose ? W[static_cast<size_t>(j) * ld + orow] : W[static_cast<size_t>(orow) * ld + j]; };
  float m = 0.f;
  for (int j = tid; j < ocol_n; j += 256) m = fmaxf(m, fabsf(at(j)));
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int se = scale_exp(m);
  const float s = pow2f(se);
  for (int j = 2 * tid; j < ocol_n; j += 512) {
    const Split2 sp = split_pair(at(j) * s, at(j + 1) * s);
    *reinterpret_cast<unsigned*>(planes + static_cast<size_t>(orow) * ocol_n + j) = sp.hi;
    *reinterpret_cast<unsigned*>(planes + n + static_cast<size_t>(orow) * ocol_n + j) = sp.lo;
  }
  if (tid == 0) winv[orow] = pow2f(-se);
}

template <bool ACC, bool RELU, bool IL>
__global__ __launch_bounds__(NT) void gemm_nt_h2w8_kernel(const float* __restrict__ X, int ldx, int M, int K,
                                                          const unsigned short* __restrict__ Bp, int N, const float* __restrict__ winv,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                          int row_blocks, int col_blocks, const float* __restrict__ X2, int ldx2,
                                                          int ns1, const float* __restrict__ rm1, const float* __restrict__ rm2,
                                                          float* __restrict__ rm2_out) {
  // buffer b: X planes [2][256][4] then B planes [2][128][4]; the accumulating instantiations park the 256 x 128 output tile here (132 KB)
  constexpr int kTileChunks = BM8 * (BN + 4) / 4;
#ifndef UAVGNN_GEMM_H2_TILE_STORE
#define UAVGNN_GEMM_H2_TILE_STORE 1   /* 0: the non-accumulating instantiations store straight from the D layout (64 dword stores per lane) */
#endif
  constexpr bool kTileStore = ACC || UAVGNN_GEMM_H2_TILE_STORE;
  __shared__ u32x4 smem[(kTileStore && kTileChunks > 2 * BUF) ? kTileChunks : 2 * BUF];
  __shared__ float sInv[BM8];       // 2^-e of the block's rows
  __shared__ float sRow2[BM8];      // rm2_out: the row maxima of X2 this workgroup computed itself
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
  float sca[4];
  auto bound = [&](int row) {   // the row's bound: the larger of the (up to two) producers'
    const float a = rm1[row];
    return rm2 != nullptr ? fmaxf(a, rm2[row]) : a;
  };
  const bool own2 = rm2_out != nullptr;   // the second source has no producer that bounds it: this launch takes its row maxima itself
  float m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + lr + 64 * i, M - 1);
    xo[i] = static_cast<unsigned>(row) * ldx + 4 * c4;
    xo2[i] = static_cast<unsigned>(row) * ldx2 + 4 * c4;
  }
  if (own2) {
    // the eight threads of a row (consecutive lanes) read its K - K1 columns as float4 c4, c4 + 8, ...: 96 columns = three per thread and
    // row, requested with the first slices of X; Inf / NaN make the bound infinite (the contract of uavgnn_row_absmax).  Every column
    // block computes the same maxima (a maximum is order-independent: bit-identical); column block 0 writes them out.
    const int nq = (K - ns1 * BK) / 4;
    bool bad[4] = {false, false, false, false};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      for (int q = c4; q < nq; q += 8) {
        const float4 v = *reinterpret_cast<const float4*>(X2 + xo2[i] + 4 * (q - c4));
        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
        bad[i] |= !(a0 <= 3.4028234663852886e38f) | !(a1 <= 3.4028234663852886e38f) | !(a2 <= 3.4028234663852886e38f) |
                  !(a3 <= 3.4028234663852886e38f);
        m2[i] = fmaxf(fmaxf(m2[i], fmaxf(a0, a1)), fmaxf(a2, a3));
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (bad[i]) m2[i] = INFINITY;
      m2[i] = fmaxf(m2[i], __shfl_xor(m2[i], 1));
      m2[i] = fmaxf(m2[i], __shfl_xor(m2[i], 2));
      m2[i] = fmaxf(m2[i], __shfl_xor(m2[i], 4));
      if (c4 == 0) {
        sRow2[lr + 64 * i] = m2[i];
        if (cb == 0 && m0 + lr + 64 * i < M) rm2_out[m0 + lr + 64 * i] = m2[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + lr + 64 * i, M - 1);
    sca[i] = pow2f(scale_exp(own2 ? fmaxf(rm1[row], m2[i]) : bound(row)));
  }
  if (!own2 && tid < BM8) sInv[tid] = pow2f(-scale_exp(bound(min(m0 + tid, M - 1))));
  const int sa_w = lr * 32 + (((c4 >> 1) ^ swz32(lr)) * 8) + (c4 & 1) * 4;   // f16 units inside an X plane
  // B loader: chunk q = tid + 512 i (i < 2): plane i, row tid / 4, chunk tid % 4
  unsigned bo[2];
  int sbw[2];
  const unsigned plane = static_cast<unsigned>(N) * K;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = tid >> 2, c = tid & 3;
    bo[i] = i * plane + static_cast<unsigned>(min(n0 + row, N - 1)) * K + 8 * c;
    sbw[i] = 2 * PA + i * PB + row * 4 + (c ^ swz32(row));
  }
  const int ns = K / BK;
  float4 ra[4];
  u32x4 rw[2];
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
    for (int i = 0; i < 2; ++i) rw[i] = *reinterpret_cast<const u32x4*>(Bp + (bo[i] + k0));
  };
  auto gload = [&](int t) {
    gload_a(t);
    gload_w(t);
  };
  auto lstore_a = [&](int buf, int i) {
    unsigned short* sa = reinterpret_cast<unsigned short*>(smem + buf * BUF) + sa_w;
    stage4(sa + 64 * i * 32, PA * 8, ra[i], sca[i]);
  };
  auto lstore_b = [&](int buf) {
    u32x4* sb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 2; ++i) sb[sbw[i]] = rw[i];
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) lstore_a(buf, i);
    lstore_b(buf);
  };
  struct Half {
    f16x8 a[2][2], b[2][2];   // [tile][plane]
  };
#define UAVGNN_H2_READ(F, buf, kh)                                                                                 \
  {                                                                                                                \
    const u32x4* sb = smem + (buf) * BUF;                                                                          \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {               \
      F.a[a][pl] = as_frag(sb[pl * PA + (wm + a * 32 + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);                        \
      F.b[a][pl] = as_frag(sb[2 * PA + pl * PB + (wn + a * 32 + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);               \
    }                                                                                                              \
  }
#define UAVGNN_H2_TERM(ia, ib)                                                                      \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)      \
      acc[a][b] = mfma32(F.a[a][ia], F.b[b][ib], acc[a][b]);
#define UAVGNN_H2_MFMA(F_)                                            \
  {                                                                   \
    const Half& F = F_;                                               \
    UAVGNN_H2_TERM(0, 1) UAVGNN_H2_TERM(1, 0) UAVGNN_H2_TERM(0, 0)    \
  }
  gload(0);
  lstore(0);
  gload(min(1, ns - 1));
  __syncthreads();
  if (own2 && tid < BM8) sInv[tid] = pow2f(-scale_exp(fmaxf(rm1[min(m0 + tid, M - 1)], sRow2[tid])));   // (read behind the loop's barriers)
  const bool early = wave < 4;
  Half f0, f1;
  UAVGNN_H2_READ(f0, 0, 0)
  if (IL) {
    for (int t = 0; t < ns; ++t) {
      UAVGNN_H2_READ(f1, t & 1, 1)
      __builtin_amdgcn_sched_barrier(0);
      const Half& F = f0;
      const int tn = min(t + 2, ns - 1);
      lstore_b((t + 1) & 1);
      gload_w(tn);
      UAVGNN_H2_TERM(0, 1)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) lstore_a((t + 1) & 1, i);
      UAVGNN_H2_TERM(1, 0)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      gload_a(tn);
      UAVGNN_H2_TERM(0, 0)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      UAVGNN_H2_READ(f0, (t + 1) & 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_H2_MFMA(f1)
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int t = 0; t < ns; ++t) {
      UAVGNN_H2_READ(f1, t & 1, 1)
      if (early) {
        lstore((t + 1) & 1);                 // slice t + 1 (the tail re-stages the last slice: unconditional, straight-line)
        gload(min(t + 2, ns - 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_H2_MFMA(f0)
      __builtin_amdgcn_sched_barrier(0);
      if (!early) {
        lstore((t + 1) & 1);
        gload(min(t + 2, ns - 1));
      }
      __syncthreads();
      UAVGNN_H2_READ(f0, (t + 1) & 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_H2_MFMA(f1)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef UAVGNN_H2_MFMA
#undef UAVGNN_H2_TERM
#undef UAVGNN_H2_READ
  // un-scaling: two exact power-of-two factors per element (2^-e_col first: weights are small, it cannot overflow; then 2^-e_row)
  if constexpr (kTileStore) {
  if ((ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && n0 + BN <= N) {
    // Y (+)= product through an LDS tile (see gemm_x3.hip: row-contiguous float4 accesses - 16 per lane instead of 64 dword stores
    // straight from the D layout; accumulating: a read-modify-write with one exposed round trip per tile)
    constexpr int LDT = BN + 4;
    float* sT = reinterpret_cast<float*>(smem);           // [256][LDT] fp32
    static_assert(BM8 * LDT * 4 <= static_cast<int>(sizeof(smem)), "the output tile fits the LDS array");
    __syncthreads();                                     // the last fragment reads of the stale buffer
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn + b * 32 + l32;
        const float bv = bias != nullptr ? bias[col] : 0.f, ci = winv[col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int lrow = wm + a * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
          sT[lrow * LDT + wn + b * 32 + l32] = acc[a][b][i] * ci * sInv[lrow] + bv;
        }
      }
    float4 yv[16];
    if constexpr (ACC) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + NT * q, row = idx >> 5, cc = idx & 31;
        yv[q] = *reinterpret_cast<const float4*>(Y + static_cast<size_t>(min(m0 + row, M - 1)) * ldy + n0 + 4 * cc);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = tid + NT * q, row = idx >> 5, cc = idx & 31;
      const float4 t = *reinterpret_cast<const float4*>(sT + row * LDT + 4 * cc);
      float4 o = t;
      if constexpr (ACC) o = {yv[q].x + t.x, yv[q].y + t.y, yv[q].z + t.z, yv[q].w + t.w};
      if (RELU) o = {fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
      if (m0 + row < M) *reinterpret_cast<float4*>(Y + static_cast<size_t>(m0 + row) * ldy + n0 + 4 * cc) = o;
    }
    return;
  }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + wn + b * 32 + l32;
    if (col >= N) continue;
    const float bv = bias != nullptr ? bias[col] : 0.f, ci = winv[col];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lrow = wm + a * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
        const int row = m0 + lrow;
        if (row < M) {
          float* p = Y + static_cast<size_t>(row) * ldy + col;
          float v = acc[a][b][i] * ci * sInv[lrow] + bv;
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

extern "C" int uavgnn_gemm_h2_supported(int M, int N, int K) {
  // 32-bit element offsets inside the kernel: every operand below 2^31 elements
  return (M > 0 && N > 0 && K >= BK && K % BK == 0 && static_cast<long long>(M) * K < (1LL << 31) &&
          2LL * N * K < (1LL << 31)) ? 1 : 0;
}

// bytes of the `planes` of a weight with `n_out` output rows and contraction length K: [2][n_out][K] f16 + n_out floats
extern "C" long long uavgnn_split_h2_bytes(int n_out, int K) {
  if (n_out <= 0 || K <= 0) return 0;
  return 4LL * n_out * K + 4LL * n_out;
}

// W [R, C] (row stride ld): transpose = 0 -> B = W (n_out = R, K = C: y = x W^T); transpose = 1 -> B = W^T (n_out = C, K = R: dx = dy W)
extern "C" int uavgnn_split_h2(const float* W, int ld, int R, int C, int transpose, void* planes, uavgnn_stream_t stream) {
  if (!W || !planes || R <= 0 || C <= 0 || ld < C) return UAVGNN_EINVAL;
  const int n_out = transpose ? C : R, K = transpose ? R : C;
  if ((K & 1) || (reinterpret_cast<uintptr_t>(planes) & 15)) return UAVGNN_EUNSUPPORTED;
  unsigned short* p = static_cast<unsigned short*>(planes);
  float* winv = reinterpret_cast<float*>(p + 2LL * n_out * K);
  hipLaunchKernelGGL(split_h2_kernel, dim3(n_out), dim3(256), 0, static_cast<hipStream_t>(stream), W, ld, R, C, transpose, p, winv);
  return launch_status();
}

// Y = [X (K1 columns) || X2 (K - K1 columns; NULL: one source, K1 = K)] B^T (+ bias) (+ Y) (ReLU) on the f16x2 arithmetic.
// rowmax / rowmax2 (the second may be NULL): per row of the activation operand an upper bound of max |.| over the row - the larger of
// the two is used - tight to within its power of two, from the kernels that produced the operand.  epilogue: UAVGNN_GEMM_ACCUMULATE,
// UAVGNN_GEMM_RELU, UAVGNN_GEMM_STAGING_INTERLEAVED of uavgnn_gemm_nt_x3.  The eight-wave kernel only (no tile variants).
static int gemm_nt_h2_launch(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const float* rowmax,
                             const float* rowmax2, float* rowmax2_out, const void* planes, int N, const float* bias, float* Y, int ldy,
                             int epilogue, uavgnn_stream_t stream) {
  if (M < 0 || !X || !planes || !Y || !rowmax || ldx < K1 || ldy < N || K1 <= 0 || K1 > K || (X2 == nullptr) != (K1 == K) ||
      (X2 != nullptr && ldx2 < K - K1) || (rowmax2_out != nullptr && (X2 == nullptr || rowmax2 != nullptr)))
    return UAVGNN_EINVAL;
  if (M == 0) return 0;
  if (!uavgnn_gemm_h2_supported(M, N, K) || (ldx & 3) || static_cast<long long>(M) * ldx >= (1LL << 31) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(planes)) & 15) ||
      (epilogue & (UAVGNN_GEMM_TILE_128 | UAVGNN_GEMM_TILE_64)))
    return UAVGNN_EUNSUPPORTED;
  if (X2 != nullptr && ((K1 % BK) || (ldx2 & 3) || static_cast<long long>(M) * ldx2 >= (1LL << 31) || (reinterpret_cast<uintptr_t>(X2) & 15)))
    return UAVGNN_EUNSUPPORTED;
  const int ns1 = K1 / BK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short* bp = static_cast<const unsigned short*>(planes);
  const float* winv = reinterpret_cast<const float*>(bp + 2LL * N * K);
  const bool acc = (epilogue & UAVGNN_GEMM_ACCUMULATE) != 0, relu = (epilogue & UAVGNN_GEMM_RELU) != 0;
  const int col_blocks = (N + BN - 1) / BN, row_blocks = (M + BM8 - 1) / BM8;
  const dim3 grid(((row_blocks + 7) / 8) * 8 * col_blocks), block(NT);
#define UAVGNN_H2_GEMM(ACC, RELU, IL)                                                                                              \
  hipLaunchKernelGGL((gemm_nt_h2w8_kernel<ACC, RELU, IL>), grid, block, 0, st, X, ldx, M, K, bp, N, winv, bias, Y, ldy, row_blocks, \
                     col_blocks, X2, ldx2, ns1, rowmax, rowmax2, rowmax2_out)
#define UAVGNN_H2_GEMM_IL(IL)                         \
  if (acc && relu) UAVGNN_H2_GEMM(true, true, IL);    \
  else if (acc) UAVGNN_H2_GEMM(true, false, IL);      \
  else if (relu) UAVGNN_H2_GEMM(false, true, IL);     \
  else UAVGNN_H2_GEMM(false, false, IL);
  if (epilogue & UAVGNN_GEMM_STAGING_INTERLEAVED) { UAVGNN_H2_GEMM_IL(true) }
  else { UAVGNN_H2_GEMM_IL(false) }
#undef UAVGNN_H2_GEMM_IL
#undef UAVGNN_H2_GEMM
  return launch_status();
}

extern "C" int uavgnn_gemm_nt_h2(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const float* rowmax,
                                 const float* rowmax2, const void* planes, int N, const float* bias, float* Y, int ldy, int epilogue,
                                 uavgnn_stream_t stream) {
  return gemm_nt_h2_launch(X, ldx, K1, X2, ldx2, M, K, rowmax, rowmax2, nullptr, planes, N, bias, Y, ldy, epilogue, stream);
}

// ... for a second source WITHOUT a producer that bounds its rows: the launch takes the row maxima of X2 [M, K - K1] itself (its workgroups
// read their 256 rows of X2 once more in front of the first slice - 96 columns at the recurrent step: a microsecond against the 8-us pass of
// uavgnn_row_absmax) and writes them to rowmax2_out [M] (Inf for a row that holds Inf / NaN; what uavgnn_row_absmax(X2) returns, bit for bit).
extern "C" int uavgnn_gemm_nt_h2_rm2(const float* X, int ldx, int K1, const float* X2, int ldx2, int M, int K, const float* rowmax,
                                     float* rowmax2_out, const void* planes, int N, const float* bias, float* Y, int ldy, int epilogue,
                                     uavgnn_stream_t stream) {
  if (!rowmax2_out) return UAVGNN_EINVAL;
  return gemm_nt_h2_launch(X, ldx, K1, X2, ldx2, M, K, rowmax, nullptr, rowmax2_out, planes, N, bias, Y, ldy, epilogue, stream);
}
