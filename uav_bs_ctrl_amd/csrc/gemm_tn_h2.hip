// Weight gradients of the dense layers on the f16 matrix cores with exactly scaled two-term splits ("f16x2", csrc/gru_h2.hip):
//   P[s][Mo, Ko] (+)= dY[rows_s, :Mo]^T X[rows_s, :Ko],   fp32 in / out / accumulate, three f16 x f16 MFMA products per fp32 product.
// Replaces autograd's dW = dy^T x of the nn.GRUCell / nn.Linear layers of /root/reference/algos/madrqn/agents/gnn_agents.py:246 (f_udt),
// :99 (f_aggr) under loss.backward() (algos/madrqn/learner.py:157) over the time-batched rows of a BPTT sequence (1.67 M at C3), which
// rounds 2-5 ran on the vendor's batched split-K fp32 GEMM (fp32 MFMA: 1/16 of the 16-bit matrix-core rate; 129-135 TFLOP/s).
//
// The contraction runs over the ROW index of both operands, so
//   * the power-of-two scales of the split are per COLUMN (per output feature of dY, per input feature of X): the column maxima come
//     from the caller (uavgnn_col_absmax, or a producer that tracks them); an element below 2^-17 of its column's maximum loses the low
//     bits of its lo term - absolute error <= 2^-39 x (column maximum), summed over 10^6 rows still ~10^-9 of a typical entry;
//   * both operands are row-major with the contraction index as the SLOW index - the opposite of what an MFMA fragment wants (8
//     consecutive k per lane).  csrc/gemm_tn_x3.hip transposes on the way INTO LDS (one dword load per element and lane: 82-98
//     TFLOP/s).  Here the tiles are staged as they come - float4 loads of whole rows, scaled, split, 8-byte LDS writes - into [16 columns]
//     x [32 rows] sub-tiles, and the fragments are read with gfx950's transposing LDS read (ds_read_b64_tr_b16: the 16 lanes of a group
//     name a 4-row x 16-column block, lane c receives column c's four k values): two of them per fragment, no VALU.
// One workgroup: 256 x 128 output tile, eight wavefronts of 64 x 64 (2 x 2 tiles of v_mfma_f32_32x32x16_f16; the 16-column sub-tiles of a
// wavefront are interleaved over the tile so that the transposing reads are free of bank conflicts - see `fo`), LDS double-buffered with
// ONE barrier per 32-row slice, the loads of slice t + 2 in flight while slice t computes (the schedule of csrc/gemm_h2.hip); the rows are
// cut into S chunks, all tiles of a chunk on one XCD (a row slice is fetched from HBM once per chunk), one partial product per chunk (the
// caller sums the S partials in a fixed order: deterministic).
#include "common.h"

namespace uavgnn {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TI = 256, TJ = 128, BK = 32, NT = 512;
#ifndef GEMM_TN_H2_SUBE
#define GEMM_TN_H2_SUBE 528   // 2-byte elements per [32 rows][16 columns] sub-tile: 512 + 16 (adjacent sub-tiles 8 banks apart: the 8-byte writes of a wavefront - 16 sub-tiles x 32 bytes - are conflict-free)
#endif
constexpr int SUBE = GEMM_TN_H2_SUBE;
constexpr int PA = (TI / 16) * SUBE, PB = (TJ / 16) * SUBE;   // elements per plane of the dY / X tile
constexpr int BUF = 2 * PA + 2 * PB;                           // elements per buffer

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
// four consecutive columns of one row, each with its own column scale -> one 8-byte group per plane
__device__ __forceinline__ void stage4(unsigned short* p, int plane_stride, float4 v, float4 s) {
  const Split2 a = split_pair(v.x * s.x, v.y * s.y), b = split_pair(v.z * s.z, v.w * s.w);
  *reinterpret_cast<u32x2*>(p) = u32x2{a.hi, b.hi};
  *reinterpret_cast<u32x2*>(p + plane_stride) = u32x2{a.lo, b.lo};
}
// one MFMA operand fragment (8 consecutive k of the lane's column) from a [k][column] sub-tile image: two transposing reads
__device__ __forceinline__ f16x8 tr_frag(const unsigned short* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));            // (generic -> LDS address space: a C-style cast)
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 16));
  return __builtin_bit_cast(f16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// column maxima: out[c] = max over rows of |x[r][c]| (bit patterns of non-negative floats order like integers: atomic integer max;
// `out` zeroed by the caller; Inf for a column that holds Inf / NaN)
__global__ __launch_bounds__(256) void col_absmax_kernel(const float* __restrict__ x, long long ld, long long n, int C,
                                                         unsigned* __restrict__ out) {
  __shared__ unsigned red[4][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.y * 64 + lane) * 4;
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  bool bad = false;
  if (c < C) {
    const int cc = min(c, C - 4);
    for (long long r = static_cast<long long>(blockIdx.x) * 4 + wave; r < n; r += static_cast<long long>(gridDim.x) * 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + cc);
      const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bad |= !(a[t] <= 3.4028234663852886e38f);
        m[t] = fmaxf(m[t], a[t]);
      }
    }
    if (bad) m[0] = m[1] = m[2] = m[3] = INFINITY;   // (conservative: the four columns of the lane)
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) red[wave][lane * 4 + t] = __float_as_uint(m[t]);
  __syncthreads();
  if (wave == 0 && c < C) {
    const int cc = min(c, C - 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned v = max(max(red[0][lane * 4 + t], red[1][lane * 4 + t]), max(red[2][lane * 4 + t], red[3][lane * 4 + t]));
      atomicMax(out + cc + t, v);
    }
  }
}

template <bool ACC>
__global__ __launch_bounds__(NT) void gemm_tn_h2_kernel(const float* __restrict__ Yd, int ldy, int Mo, const float* __restrict__ X,
                                                        int ldx, int Ko, long long n_rows, long long chunk,
                                                        const float* __restrict__ cmax_y, const float* __restrict__ cmax_x,
                                                        float* __restrict__ P, int col_blocks, int tiles, int S_all) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];   // buffer b: dY planes [2][16 sub-tiles] then X planes [2][8 sub-tiles]
  __shared__ float sInvA[TI], sInvB[TJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> (output tile, row chunk): consecutive ids go to consecutive XCDs; all tiles of chunk s sit on XCD s % 8
  int tile, s;
  {
    const int id = blockIdx.x;
    if ((S_all & 7) == 0) {
      const int xcd = id & 7, k = id >> 3;
      tile = k % tiles;
      s = (k / tiles) * 8 + xcd;
    } else {
      tile = id % tiles;
      s = id / tiles;
    }
  }
  const int rb = tile / col_blocks, cb = tile - rb * col_blocks;
  const int m0 = rb * TI, n0 = cb * TJ;
  const long long r_begin = static_cast<long long>(s) * chunk, r_end = min(r_begin + chunk, n_rows);
  const int ns = r_begin < r_end ? static_cast<int>((r_end - r_begin) / BK) : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // loaders: dY tile [32 rows][256 columns]: thread -> columns 4 (tid % 64) .. + 3, rows tid / 64 + 8 q (q < 4);
  //          X  tile [32 rows][128 columns]: thread -> columns 4 (tid % 32) .. + 3, rows tid / 32 + 16 q (q < 2).
  // Columns past Mo / Ko are clamped to the last four (their outputs are never stored).
  const int ca = min(m0 + 4 * (tid & 63), Mo - 4), cx = min(n0 + 4 * (tid & 31), Ko - 4);
  const int ka = tid >> 6, kx = tid >> 5;
  float4 sca, scx;
  {
    const float4 ma = *reinterpret_cast<const float4*>(cmax_y + ca), mx = *reinterpret_cast<const float4*>(cmax_x + cx);
    sca = make_float4(pow2f(scale_exp(ma.x)), pow2f(scale_exp(ma.y)), pow2f(scale_exp(ma.z)), pow2f(scale_exp(ma.w)));
    scx = make_float4(pow2f(scale_exp(mx.x)), pow2f(scale_exp(mx.y)), pow2f(scale_exp(mx.z)), pow2f(scale_exp(mx.w)));
  }
  if (tid < TI) sInvA[tid] = pow2f(-scale_exp(cmax_y[min(m0 + tid, Mo - 1)]));
  if (tid < TJ) sInvB[tid] = pow2f(-scale_exp(cmax_x[min(n0 + tid, Ko - 1)]));
  const float* pa = Yd + (r_begin + ka) * ldy + ca;
  const float* px = X + (r_begin + kx) * ldx + cx;
  const long long step_a = 8LL * ldy, step_x = 16LL * ldx;
  // LDS element offsets of the thread's 8-byte groups inside a plane: sub-tile (column / 16), row k, column % 16
  const int wa = ((tid & 63) >> 2) * SUBE + ka * 16 + 4 * (tid & 3);
  const int wx = ((tid & 31) >> 2) * SUBE + kx * 16 + 4 * (tid & 3);
  float4 ra[4], rx[2];
  auto gload = [&](int t) {   // slice t (clamped to the chunk's last slice: the tail re-reads it, static vmcnt)
    const long long off = static_cast<long long>(min(t, ns - 1)) * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(pa + off * ldy + q * step_a);
#pragma unroll
    for (int q = 0; q < 2; ++q) rx[q] = *reinterpret_cast<const float4*>(px + off * ldx + q * step_x);
  };
  auto lstore = [&](int buf) {
    unsigned short* sb = smem + buf * BUF;
#pragma unroll
    for (int q = 0; q < 4; ++q) stage4(sb + wa + q * 8 * 16, PA, ra[q], sca);
#pragma unroll
    for (int q = 0; q < 2; ++q) stage4(sb + 2 * PA + wx + q * 16 * 16, PB, rx[q], scx);
  };
  // fragment addresses: lane l of a 16-lane group names row (l & 15) / 4 and columns 4 (l & 3) .. + 3 of a 4 x 16 block; the group is
  // half ((l >> 4) & 1) of the 32-column MFMA tile and k group (l >> 5).  The two halves of an MFMA tile are sub-tiles FOUR apart: the
  // transposing read serves lanes 0-31 in one LDS cycle only if their two 128-byte blocks fall on disjoint halves of the 64 banks, and
  // 4 x 264 dwords = 32 banks (mod 64), where adjacent sub-tiles - 8 banks apart, the spacing the 8-byte WRITES need - overlap on 24 of
  // 32 banks (measured before this: SQ_LDS_BANK_CONFLICT = 36 % of the kernel's LDS cycles).  So a wavefront's 64 x 64 output is not a
  // contiguous block: its dY columns are sub-tiles {wr + 8 a + 4 half}, its X columns {wc + 2 b + 4 half} (wr = wave / 2, wc = wave % 2,
  // a / b = MFMA tile); the epilogue stores by the same map.
  const int fo = ((lane >> 4) & 1) * 4 * SUBE + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 16 + 4 * (lane & 3);
  const int wr = wave >> 1, wcx = wave & 1;
  struct Half {
    f16x8 a[2][2], b[2][2];   // [tile][plane]
  };
#define UAVGNN_TNH2_READ(F, buf, kh)                                                                          \
  {                                                                                                           \
    const unsigned short* sb = smem + (buf) * BUF + fo + (kh) * 16 * 16;                                      \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {          \
      F.a[a][pl] = tr_frag(sb + pl * PA + (wr + 8 * a) * SUBE);                                               \
      F.b[a][pl] = tr_frag(sb + 2 * PA + pl * PB + (wcx + 2 * a) * SUBE);                                     \
    }                                                                                                         \
  }
#define UAVGNN_TNH2_TERM(ia, ib)                                                                    \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)      \
      acc[a][b] = mfma32(F.a[a][ia], F.b[b][ib], acc[a][b]);
#define UAVGNN_TNH2_MFMA(F_)                                                \
  {                                                                         \
    const Half& F = F_;                                                     \
    UAVGNN_TNH2_TERM(0, 1) UAVGNN_TNH2_TERM(1, 0) UAVGNN_TNH2_TERM(0, 0)    \
  }
  if (ns > 0) {
    gload(0);
    lstore(0);
    gload(1);
    __syncthreads();
    const bool early = wave < 4;
    Half f0, f1;
    UAVGNN_TNH2_READ(f0, 0, 0)
    for (int t = 0; t < ns; ++t) {
      UAVGNN_TNH2_READ(f1, t & 1, 1)
      if (early) {
        lstore((t + 1) & 1);
        gload(t + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_TNH2_MFMA(f0)
      __builtin_amdgcn_sched_barrier(0);
      if (!early) {
        lstore((t + 1) & 1);
        gload(t + 2);
      }
      __syncthreads();
      UAVGNN_TNH2_READ(f0, (t + 1) & 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      UAVGNN_TNH2_MFMA(f1)
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    __syncthreads();   // sInvA / sInvB
  }
#undef UAVGNN_TNH2_MFMA
#undef UAVGNN_TNH2_TERM
#undef UAVGNN_TNH2_READ
  // D layout of a 32 x 32 tile: lane l holds column l % 32, register i holds row 8 (i / 4) + 4 (l / 32) + i % 4
  float* __restrict__ Ps = P + static_cast<size_t>(s) * Mo * Ko;
  const int l32 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int lcol = 16 * (wcx + 2 * b + 4 * (l32 >> 4)) + (l32 & 15), col = n0 + lcol;
    if (col >= Ko) continue;
    const float cj = sInvB[lcol];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int i32 = 8 * (i >> 2) + 4 * lh + (i & 3);                                   // the MFMA tile's row
        const int lrow = 16 * (wr + 8 * a + 4 * (i32 >> 4)) + (i32 & 15), row = m0 + lrow;
        if (row < Mo) {
          float* p = Ps + static_cast<size_t>(row) * Ko + col;
          const float v = acc[a][b][i] * cj * sInvA[lrow];      // two exact power-of-two factors
          *p = ACC ? *p + v : v;
        }
      }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

// out [C] = max over the n rows of |x[r][c]| (Inf for a column that holds Inf / NaN): the column scales of uavgnn_gemm_tn_h2.  C % 4 == 0,
// 16-byte aligned rows.  Deterministic (a maximum).
extern "C" int uavgnn_col_absmax(const float* x, long long ld, long long n, int C, float* out, uavgnn_stream_t stream) {
  if (!x || !out || n <= 0 || C <= 0 || ld < C) return UAVGNN_EINVAL;
  if ((C & 3) || (ld & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return UAVGNN_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, sizeof(float) * C, st) != hipSuccess) return launch_status();
  const int gy = (C + 255) / 256;
  long long gx = (n + 63) / 64;
  if (gx > 2048 / gy) gx = 2048 / gy;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(col_absmax_kernel, dim3(static_cast<unsigned>(gx), gy), dim3(256), 0, st, x, ld, n, C,
                     reinterpret_cast<unsigned*>(out));
  return launch_status();
}

extern "C" int uavgnn_gemm_tn_h2_supported(long long n_rows, int Mo, int Ko) {
  return (n_rows > 0 && (n_rows % BK) == 0 && Mo >= 4 && Ko >= 4 && (Mo % 4) == 0 && (Ko % 4) == 0) ? 1 : 0;
}

extern "C" int uavgnn_gemm_tn_h2_chunks(long long n_rows, int Mo, int Ko) {
  if (!uavgnn_gemm_tn_h2_supported(n_rows, Mo, Ko)) return 0;
  const int tiles = ((Mo + TI - 1) / TI) * ((Ko + TJ - 1) / TJ);
  // One workgroup per CU (101 KB of LDS), every workgroup the same length: the grid is cut for TWO FULL rounds of the 256 CUs - the largest
  // chunk count, in whole rounds over the 8 XCDs, with tiles x S <= 512.  (Rounding S UP - 9 tiles x 64 = 576, 6 x 88 = 528 workgroups - left
  // a third round for 16 .. 64 workgroups: dW_ih 4.45 -> 3.9 ms, dW_hh 3.34 -> 2.9 ms with the cut below.)
  long long S = 512 / tiles;
  if (S >= 8) S = S / 8 * 8;
  if (S < 1) S = 1;
  const long long max_s = n_rows / 512 > 0 ? n_rows / 512 : 1;     // chunks of at least 512 rows
  if (S > max_s) S = max_s;
  return static_cast<int>(S);
}

// partials [S][Mo][Ko] (+)= dY[rows of chunk s, :Mo]^T X[rows of chunk s, :Ko] on the f16x2 arithmetic.  colmax_y [Mo] / colmax_x [Ko]:
// upper bounds of max |.| over every column of dY / X, tight to within their power of two (uavgnn_col_absmax).  n_rows % 32 == 0, Mo and Ko
// multiples of 4, 16-byte aligned rows (UAVGNN_EUNSUPPORTED otherwise); S = uavgnn_gemm_tn_h2_chunks or any positive count.
extern "C" int uavgnn_gemm_tn_h2(const float* dY, long long ldy, int Mo, const float* X, long long ldx, int Ko, long long n_rows,
                                 const float* colmax_y, const float* colmax_x, float* partials, int S, int accumulate,
                                 uavgnn_stream_t stream) {
  if (!dY || !X || !partials || !colmax_y || !colmax_x || Mo <= 0 || Ko <= 0 || n_rows <= 0 || S <= 0 || ldy < Mo || ldx < Ko)
    return UAVGNN_EINVAL;
  if (!uavgnn_gemm_tn_h2_supported(n_rows, Mo, Ko) || (ldy & 3) || (ldx & 3) || ldy >= (1LL << 31) || ldx >= (1LL << 31) ||
      ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(colmax_y) |
        reinterpret_cast<uintptr_t>(colmax_x)) & 15) || (reinterpret_cast<uintptr_t>(partials) & 3))
    return UAVGNN_EUNSUPPORTED;
  long long chunk = (n_rows + S - 1) / S;
  chunk = (chunk + BK - 1) / BK * BK;                    // whole 32-row slices; the last chunk may be shorter (or empty)
  const int col_blocks = (Ko + TJ - 1) / TJ, row_blocks = (Mo + TI - 1) / TI, tiles = row_blocks * col_blocks;
  const dim3 grid(static_cast<unsigned>(tiles) * S), block(NT);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (accumulate)
    hipLaunchKernelGGL(gemm_tn_h2_kernel<true>, grid, block, 0, st, dY, static_cast<int>(ldy), Mo, X, static_cast<int>(ldx), Ko, n_rows,
                       chunk, colmax_y, colmax_x, partials, col_blocks, tiles, S);
  else
    hipLaunchKernelGGL(gemm_tn_h2_kernel<false>, grid, block, 0, st, dY, static_cast<int>(ldy), Mo, X, static_cast<int>(ldx), Ko, n_rows,
                       chunk, colmax_y, colmax_x, partials, col_blocks, tiles, S);
  return launch_status();
}
