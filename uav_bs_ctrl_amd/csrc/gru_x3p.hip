// K4 from PREPARED operand planes: the GRU cell of csrc/gru_x3.hip (nn.GRUCell at
// /root/reference/algos/madrqn/agents/gnn_agents.py:246, gate order r, z, n; same tile, same six-product bf16x3 arithmetic, same
// accumulation order: results are BIT-IDENTICAL to uavgnn_gru_cell_fwd_x3) with the in-kernel operand split removed.
//
// csrc/gru_x3.hip loads its activations as fp32 and splits them into three bf16 planes while it stages them - ~70 VALU + 35 LDS
// stores + 7 loads per 36 MFMAs of a wavefront, repeated by each of the four column-block workgroups of a row block; every
// non-MFMA instruction a SIMD issues costs ~4 cycles of matrix-core issue, which bounds that kernel at ~72 %
// of its MFMA time.  Here the split exists already:
//   * activations: uavgnn_tarmac_msg_fwd (csrc/tarmac_msg.hip) wrote [x || c || h] as bf16 planes in THIS kernel's tile order -
//     per (row block of 128, K slice of 32) three planes of 128 rows x 4 sixteen-byte chunks, chunk g of row r at r * 4 + (g ^
//     swz32(r)): the LDS image of csrc/gru_x3.hip, verbatim;
//   * weights: uavgnn_gru_split_weight_tiles lays [W_ih | W_hh] out per (column block of 64 units, K slice) as three planes of
//     192 rows (gate * 64 + unit) in the same chunk order.
// Staging a slice is therefore a LINEAR copy of 24 + 36 KB, done by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction,
// no VGPR, no VALU, no ds_write): 7-8 instructions per wavefront per slice beside its 36 MFMAs and 24 fragment reads.
// Double-buffered LDS, one barrier per slice; the DMA of slice t + 1 is issued at the top of iteration t and drained by the
// `s_waitcnt vmcnt(0)` of the barrier that ends it.
#include "bf16x3.h"
#include "common.h"
#include "uavgnn_probe.h"

namespace uavgnn {
namespace {

using namespace x3;
constexpr int BM = 128, BK = 32, BJ = 64, NT = 512, ST = 68;
constexpr int PA = BM * 4, PB = 3 * BJ * 4;            // 16-byte chunks per split plane of the A / B tile

__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

// 64 consecutive 16-byte chunks: global (per lane) -> LDS (wave-uniform base + 16 lane)
__device__ __forceinline__ void glds16(const u32x4* g, u32x4* l) {
  __builtin_amdgcn_global_load_lds(
      reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(g)),
      reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(l))), 16, 0, 0);
}

// [W_ih (K_in columns) | W_hh (H columns)] -> tiles [H / 64 column blocks][(K_in + H) / 32 slices][3 planes][192 rows][4 chunks]
__global__ __launch_bounds__(256) void gru_weight_tiles_kernel(const float* __restrict__ W_ih, int K_in, const float* __restrict__ W_hh,
                                                               int H, u32x4* __restrict__ tiles) {
  const int ns = (K_in + H) / BK, n1 = K_in / BK;
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const long long total = static_cast<long long>(H / BJ) * ns * (PB / 1);
  if (idx >= total) return;
  const int c = static_cast<int>(idx & 3), row = static_cast<int>((idx >> 2) % (3 * BJ));
  const long long rest = (idx >> 2) / (3 * BJ);
  const int s = static_cast<int>(rest % ns), cb = static_cast<int>(rest / ns);
  const int gate = row / BJ, unit = row - gate * BJ;
  const float* src = s < n1 ? W_ih + static_cast<size_t>(gate * H + cb * BJ + unit) * K_in + BK * s + 8 * c
                            : W_hh + static_cast<size_t>(gate * H + cb * BJ + unit) * H + BK * (s - n1) + 8 * c;
  const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
  const Split3 a = split_pair(lo.x, lo.y), b = split_pair(lo.z, lo.w), cc = split_pair(hi.x, hi.y), d = split_pair(hi.z, hi.w);
  u32x4* dst = tiles + (static_cast<size_t>(cb) * ns + s) * (3 * PB) + row * 4 + (c ^ swz32(row));
  dst[0] = u32x4{a.h1, b.h1, cc.h1, d.h1};
  dst[PB] = u32x4{a.h2, b.h2, cc.h2, d.h2};
  dst[2 * PB] = u32x4{a.h3, b.h3, cc.h3, d.h3};
}

#define UAVGNN_X3_FOR_TERMS(M) M(0, 2) M(2, 0) M(1, 1) M(0, 1) M(1, 0) M(0, 0)

// Timing ablations of tools/cell_ablate.py (a probe library compiled with -DUAVGNN_X3P_DBG=bits; the shipped build has 0 and
// none of this): 1 no DMA behind the first two slices (16: no activation DMA, 32: no weight DMA), 2 no gate epilogue (the accumulators are stored raw), 4 no MFMA, 8 no
// fragment reads inside the loop.  Results are garbage for every non-zero value.
#ifndef UAVGNN_X3P_DBG
#define UAVGNN_X3P_DBG 0
#endif
constexpr int DBG = UAVGNN_X3P_DBG;
__device__ __forceinline__ f32x16 mfma_dbg(bf16x8 a, bf16x8 b, f32x16 c) {
  if (DBG & 4) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
  }
  return mfma32(a, b, c);
}

// OPT (timing experiments of tools/cell_probe.py; results are bit-identical for every value):
//   bit 0  the fragment reads of the second half are issued BEHIND the first six MFMAs of the first half (the compiler waits for
//          every LDS read in flight - lgkmcnt(0) - in front of the first MFMA that follows a read in program order);
//   bit 1  THREE activation buffers: the DMA of the A planes runs two slices ahead (they come from HBM; the weight tiles are
//          L2-resident and stay one slice ahead), raw s_barrier + counted s_waitcnt vmcnt(3) instead of __syncthreads();
//   bit 2  the epilogue's h tile is requested at the top of the last slice instead of behind the loop;
//   bit 3  (without bit 1) the DMA of slice t + 2 is issued right BEHIND the barrier of iteration t - the buffer of slice t is free
//          there - instead of at the top of iteration t + 1: a full slice of MFMA work covers it, not half of one.
template <bool SAVE, int OPT>
__global__ __launch_bounds__(NT) void gru_cell_fwd_planes_kernel(const u32x4* __restrict__ Ap, const float* __restrict__ h, int N,
                                                                int H, int n12, int ns, const u32x4* __restrict__ Wt,
                                                                const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                                float* __restrict__ h_out, float* __restrict__ pre, int row_blocks) {
  constexpr bool LATE_READ = OPT & 1, DEEP = OPT & 2, EARLY_H = OPT & 4, POST = OPT & 8;
  static_assert(!(POST && DEEP), "bit 3 replaces bit 1");
  constexpr int NA_BUF = DEEP ? 3 : 2;
  constexpr int A_WORDS = 3 * PA, B_WORDS = 3 * PB;
  __shared__ u32x4 smem[NA_BUF * A_WORDS + 2 * B_WORDS];   // A planes [buf][3][128][4], then B planes [buf][3][192 = gate * 64 + unit][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int CB = H / BJ;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / CB) * 8 + xcd, cb = slot - (slot / CB) * CB;
  if (rb >= row_blocks) return;
  const int m0 = rb * BM, j0 = cb * BJ;

  f32x16 acc[4];     // 32 x 32 tile per set: r, z, gi_n, gh_n
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;
  const int l32 = lane & 31, lh = lane >> 5, sw = swz32(l32);

  const u32x4* __restrict__ a_src = Ap + static_cast<size_t>(rb) * ns * A_WORDS + lane;
  const u32x4* __restrict__ w_src = Wt + static_cast<size_t>(cb) * ns * B_WORDS + lane;
  u32x4* const sA = smem;
  u32x4* const sB = smem + NA_BUF * A_WORDS;
  // wave w copies the 1-KB pieces w, w + 8, ... of a slice's 24 (A) / 36 (B) pieces
  auto issue_a = [&](int s, int buf) {
    if ((DBG & 16) && s > 1) return;
    const u32x4* ga = a_src + static_cast<size_t>(s) * A_WORDS;
#pragma unroll
    for (int k = 0; k < 3; ++k) glds16(ga + (wave + 8 * k) * 64, sA + buf * A_WORDS + (wave + 8 * k) * 64);
  };
  auto issue_b = [&](int s, int buf) {
    if ((DBG & 32) && s > 1) return;
    const u32x4* gw = w_src + static_cast<size_t>(s) * B_WORDS;
#pragma unroll
    for (int k = 0; k < 4; ++k) glds16(gw + (wave + 8 * k) * 64, sB + buf * B_WORDS + (wave + 8 * k) * 64);
    if (wave < 4) glds16(gw + (wave + 32) * 64, sB + buf * B_WORDS + (wave + 32) * 64);
  };

  struct Half {
    bf16x8 a[3], b[3][3];   // [plane], [gate][plane]
  };
#define UAVGNN_X3P_READ(F, abuf, bbuf, kh)                                                                         \
  {                                                                                                                \
    const u32x4* sa = sA + (abuf) * A_WORDS;                                                                       \
    const u32x4* sb = sB + (bbuf) * B_WORDS;                                                                       \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) F.a[pl] = as_frag(sa[pl * PA + (wm + l32) * 4 + ((2 * (kh) + lh) ^ sw)]); \
    _Pragma("unroll") for (int gate = 0; gate < 3; ++gate) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)        \
        F.b[gate][pl] = as_frag(sb[pl * PB + (gate * BJ + wc + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);                \
  }
#define UAVGNN_X3P_TERM(ia, ib)                    \
  acc[0] = mfma_dbg(F.a[ia], F.b[0][ib], acc[0]);  \
  acc[1] = mfma_dbg(F.a[ia], F.b[1][ib], acc[1]);  \
  acc[NSET] = mfma_dbg(F.a[ia], F.b[2][ib], acc[NSET]);
#define UAVGNN_X3P_MFMA(F_, NSET_)                 \
  {                                                \
    constexpr int NSET = NSET_;                    \
    const Half& F = F_;                            \
    UAVGNN_X3_FOR_TERMS(UAVGNN_X3P_TERM)           \
  }
#define UAVGNN_X3P_MFMA_HEAD(F_, NSET_)            \
  {                                                \
    constexpr int NSET = NSET_;                    \
    const Half& F = F_;                            \
    UAVGNN_X3P_TERM(0, 2) UAVGNN_X3P_TERM(2, 0)    \
  }
#define UAVGNN_X3P_MFMA_TAIL(F_, NSET_)            \
  {                                                \
    constexpr int NSET = NSET_;                    \
    const Half& F = F_;                            \
    UAVGNN_X3P_TERM(1, 1) UAVGNN_X3P_TERM(0, 1) UAVGNN_X3P_TERM(1, 0) UAVGNN_X3P_TERM(0, 0) \
  }
  float4 hreg0, hreg1, hreg2, hreg3;    // the epilogue's h tile (named registers: an array captured by a lambda is promoted to LDS / scratch)
#define UAVGNN_X3P_LOAD_H_ONE(q, dst)                                                                                   \
  {                                                                                                                     \
    const int idx = tid + NT * (q), row = idx >> 4, cc = idx & 15;                                                      \
    dst = *reinterpret_cast<const float4*>(h + static_cast<size_t>(min(m0 + row, N - 1)) * H + j0 + 4 * cc);            \
  }
#define UAVGNN_X3P_LOAD_H UAVGNN_X3P_LOAD_H_ONE(0, hreg0) UAVGNN_X3P_LOAD_H_ONE(1, hreg1) UAVGNN_X3P_LOAD_H_ONE(2, hreg2) UAVGNN_X3P_LOAD_H_ONE(3, hreg3)
  // end-of-slice barrier: slice t + 1 must have landed (every wave's pieces), the readers of the buffers that the next iteration
  // overwrites must be through.  DEEP: the A pieces of slice t + 2 (the three youngest DMAs of this wave) stay in flight.
  auto slice_barrier = [&](bool a_in_flight) {
    if (DEEP) {
      if (a_in_flight) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      // EXPLICIT wait for this wavefront's LDS-DMA before the barrier.  __syncthreads() does not promise it: the compiler's fence
      // covers LDS reads / writes and global stores, while a global_load_lds is waited for only where the compiler happens to order
      // a later ds_read behind it.  With the DMA issued behind the barrier (bit 3) it emitted NO vmcnt wait in the whole loop and a
      // fragment read met a 1-KB piece that had not landed about once in a thousand launches (tools/cell_race.py: one 32 x 16
      // patch of h' off by 3e-7).
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  // iteration t: DMA of the next slice(s) into the free buffers (their readers passed the barrier of iteration t - 1), fragment
  // reads software-pipelined over the two 16-wide halves as in csrc/gru_x3.hip
#define UAVGNN_X3P_STEP(NSET)                                          \
  {                                                                    \
    const int ab = DEEP ? t % 3 : (t & 1), bb = t & 1;                 \
    const int ab1 = DEEP ? (t + 1) % 3 : ((t + 1) & 1);                \
    if (EARLY_H && t == ns - 1) { UAVGNN_X3P_LOAD_H }                  \
    if (!(DBG & 1) && !POST) {                                         \
      if (t + 1 < ns) issue_b(t + 1, (t + 1) & 1);                     \
      if (DEEP) { if (t + 2 < ns) issue_a(t + 2, (t + 2) % 3); }       \
      else if (t + 1 < ns) issue_a(t + 1, (t + 1) & 1);                \
    }                                                                  \
    if (LATE_READ) {                                                   \
      __builtin_amdgcn_sched_barrier(0);                               \
      UAVGNN_X3P_MFMA_HEAD(f0, NSET)                                   \
      __builtin_amdgcn_sched_barrier(0);                               \
      if (!(DBG & 8)) UAVGNN_X3P_READ(f1, ab, bb, 1)                   \
      __builtin_amdgcn_sched_barrier(0);                               \
      UAVGNN_X3P_MFMA_TAIL(f0, NSET)                                   \
    } else {                                                           \
      if (!(DBG & 8)) UAVGNN_X3P_READ(f1, ab, bb, 1)                   \
      __builtin_amdgcn_sched_barrier(0);                               \
      UAVGNN_X3P_MFMA(f0, NSET)                                        \
    }                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                 \
    slice_barrier(t + 2 < ns);                                         \
    if (POST && !(DBG & 1) && t + 2 < ns) {                            \
      issue_b(t + 2, t & 1);                                           \
      issue_a(t + 2, t & 1);                                           \
    }                                                                  \
    if (LATE_READ) {                                                   \
      __builtin_amdgcn_sched_barrier(0);                               \
      UAVGNN_X3P_MFMA_HEAD(f1, NSET)                                   \
      __builtin_amdgcn_sched_barrier(0);                               \
      if (!(DBG & 8)) UAVGNN_X3P_READ(f0, ab1, (t + 1) & 1, 0)         \
      __builtin_amdgcn_sched_barrier(0);                               \
      UAVGNN_X3P_MFMA_TAIL(f1, NSET)                                   \
    } else {                                                           \
      if (!(DBG & 8)) UAVGNN_X3P_READ(f0, ab1, (t + 1) & 1, 0)         \
      __builtin_amdgcn_sched_barrier(0);                               \
      UAVGNN_X3P_MFMA(f1, NSET)                                        \
    }                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                 \
  }

  issue_b(0, 0);
  issue_a(0, 0);
  if (DEEP && ns > 1) issue_a(1, 1);
  if (POST && ns > 1) {
    issue_b(1, 1);
    issue_a(1, 1);
  }
  slice_barrier(DEEP && ns > 1);
  Half f0, f1;
  UAVGNN_X3P_READ(f0, 0, 0, 0)
  if (DBG & 8) UAVGNN_X3P_READ(f1, 0, 0, 1)
  int t = 0;
  for (; t < n12; ++t) UAVGNN_X3P_STEP(2)
  for (; t < ns; ++t) UAVGNN_X3P_STEP(3)
#undef UAVGNN_X3P_STEP
#undef UAVGNN_X3P_MFMA_TAIL
#undef UAVGNN_X3P_MFMA_HEAD
#undef UAVGNN_X3P_MFMA
#undef UAVGNN_X3P_TERM
#undef UAVGNN_X3P_READ
  __syncthreads();   // the last iteration's read of the stale buffer must not race the epilogue's tile
  if (DBG & 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + wm + 8 * (i >> 2) + 4 * lh + (i & 3);
      if (row < N) h_out[static_cast<size_t>(row) * H + j0 + wc + l32] = acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    }
    return;
  }

  // ---- epilogue (csrc/gru_x3.hip): biases, gates, h' through an LDS tile -------------------------------------------------
  float* sH = reinterpret_cast<float*>(smem);             // [128][ST] fp32 tile: h in, h' out
  if (!EARLY_H) { UAVGNN_X3P_LOAD_H }
#define UAVGNN_X3P_PUT_H(q, src)                                        \
  {                                                                     \
    const int idx = tid + NT * (q), row = idx >> 4, cc = idx & 15;      \
    *reinterpret_cast<float4*>(sH + row * ST + 4 * cc) = src;           \
  }
  UAVGNN_X3P_PUT_H(0, hreg0) UAVGNN_X3P_PUT_H(1, hreg1) UAVGNN_X3P_PUT_H(2, hreg2) UAVGNN_X3P_PUT_H(3, hreg3)
#undef UAVGNN_X3P_PUT_H
#undef UAVGNN_X3P_LOAD_H
#undef UAVGNN_X3P_LOAD_H_ONE
  __syncthreads();
  const int c = j0 + wc + l32;
  const float b_r = b_ih[c] + b_hh[c], b_z = b_ih[H + c] + b_hh[H + c], b_in = b_ih[2 * H + c], b_hn = b_hh[2 * H + c];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int lrow = wm + 8 * (i >> 2) + 4 * lh + (i & 3);
    const int row = m0 + lrow;
    const float pr = acc[0][i] + b_r, pz = acc[1][i] + b_z, gin = acc[2][i] + b_in, ghn = acc[3][i] + b_hn;
    const float rr = sigmoidf_(pr), zz = sigmoidf_(pz);
    const float nn = tanhf_(fmaf(rr, ghn, gin));
    float* hp = sH + lrow * ST + wc + l32;
    *hp = fmaf(zz, *hp - nn, nn);
    if (SAVE && row < N) {
      float* p = pre + static_cast<size_t>(row) * 4 * H + c;
      p[0] = pr;
      p[H] = pz;
      p[2 * H] = gin;
      p[3 * H] = ghn;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + NT * q, row = idx >> 4, cc = idx & 15;
    if (m0 + row < N)
      *reinterpret_cast<float4*>(h_out + static_cast<size_t>(m0 + row) * H + j0 + 4 * cc) =
          *reinterpret_cast<const float4*>(sH + row * ST + 4 * cc);
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" long long uavgnn_gru_weight_tiles_bytes(int K_in, int H) {
  if (K_in <= 0 || H <= 0) return 0;
  return static_cast<long long>(H / BJ) * ((K_in + H) / BK) * 3 * PB * 16;
}

// [W_ih [3H, K_in] | W_hh [3H, H]] (both contiguous) -> the weight tiles uavgnn_gru_cell_fwd_planes reads
extern "C" int uavgnn_gru_split_weight_tiles(const float* W_ih, int K_in, const float* W_hh, int H, void* tiles, uavgnn_stream_t stream) {
  if (!W_ih || !W_hh || !tiles || K_in <= 0 || H <= 0) return UAVGNN_EINVAL;
  if ((K_in % BK) || (H % BJ) || ((reinterpret_cast<uintptr_t>(W_ih) | reinterpret_cast<uintptr_t>(W_hh) | reinterpret_cast<uintptr_t>(tiles)) & 15))
    return UAVGNN_EUNSUPPORTED;
  const long long total = static_cast<long long>(H / BJ) * ((K_in + H) / BK) * PB;
  hipLaunchKernelGGL(gru_weight_tiles_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W_ih, K_in, W_hh, H, static_cast<u32x4*>(tiles));
  return launch_status();
}

extern "C" int uavgnn_gru_cell_fwd_planes_opts(const void* planes, int K_in, const float* h, int N, int H, const void* tiles,
                                               const float* b_ih, const float* b_hh, float* h_out, float* pre_save, int opt,
                                               uavgnn_stream_t stream) {
  if (N < 0 || !planes || !h || !tiles || !b_ih || !b_hh || !h_out || opt < 0 || opt > 15 || ((opt & 8) && (opt & 2))) return UAVGNN_EINVAL;
  if (K_in < BK || (K_in % BK) || H < BJ || (H % BJ) ||
      ((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(tiles) |
        reinterpret_cast<uintptr_t>(h_out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const int row_blocks = (N + BM - 1) / BM, rb8 = ((row_blocks + 7) / 8) * 8;
  const int n12 = K_in / BK, ns = n12 + H / BK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(rb8 * (H / BJ)), block(NT);
  const u32x4* Ap = static_cast<const u32x4*>(planes);
  const u32x4* Wt = static_cast<const u32x4*>(tiles);
#define UAVGNN_X3P_LAUNCH(OPT_)                                                                                              \
  case OPT_:                                                                                                                 \
    if (pre_save != nullptr)                                                                                                 \
      hipLaunchKernelGGL((gru_cell_fwd_planes_kernel<true, OPT_>), grid, block, 0, st, Ap, h, N, H, n12, ns, Wt, b_ih, b_hh, h_out, \
                         pre_save, row_blocks);                                                                              \
    else                                                                                                                     \
      hipLaunchKernelGGL((gru_cell_fwd_planes_kernel<false, OPT_>), grid, block, 0, st, Ap, h, N, H, n12, ns, Wt, b_ih, b_hh, h_out, \
                         pre_save, row_blocks);                                                                              \
    break;
  switch (opt) {
    UAVGNN_X3P_LAUNCH(0) UAVGNN_X3P_LAUNCH(1) UAVGNN_X3P_LAUNCH(2) UAVGNN_X3P_LAUNCH(3) UAVGNN_X3P_LAUNCH(4) UAVGNN_X3P_LAUNCH(5)
    UAVGNN_X3P_LAUNCH(6) UAVGNN_X3P_LAUNCH(7) UAVGNN_X3P_LAUNCH(8) UAVGNN_X3P_LAUNCH(9) UAVGNN_X3P_LAUNCH(12) UAVGNN_X3P_LAUNCH(13)
  }
#undef UAVGNN_X3P_LAUNCH
  return launch_status();
}

// planes: the operand [inp (K_in columns, zero-padded to whole 32-wide slices) || h] of N rows as written by uavgnn_tarmac_msg_fwd
// (uavgnn_tarmac_msg_planes_bytes); h: the same hidden state in fp32 (the convex update reads it); tiles: uavgnn_gru_split_weight_tiles.
extern "C" int uavgnn_gru_cell_fwd_planes(const void* planes, int K_in, const float* h, int N, int H, const void* tiles,
                                          const float* b_ih, const float* b_hh, float* h_out, float* pre_save, uavgnn_stream_t stream) {
  return uavgnn_gru_cell_fwd_planes_opts(planes, K_in, h, N, H, tiles, b_ih, b_hh, h_out, pre_save, 0, stream);
}
