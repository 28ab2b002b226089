// K1 forward of the WHOLE observation encoder in ONE launch: both GATv2 relations of
// /root/reference/algos/madrqn/agents/gnn_agents.py:93-96,:103-104 (`gt -seen-> agent`, F = 4 and `ubs -near-> agent`,
// F = 2; 4 heads x 64 channels; math SURVEY Appendix A.1/A.3) write the two halves of one [N, 2H] row (the th.cat of
// gnn_agents.py:106 never exists), with one constant-load prologue and one pass over the destination meta data.
//
// Round 4 layout (rounds 1-3: csrc/gatv2_hetero_pair.inc, still the fp32-MFMA build behind `phases` bit 8).  One workgroup of
// eight wavefronts per CU; every persistent wavefront runs two phases:
//
//  S  the `seen` relation of the destinations that HAVE in-edges, one destination at a time on 16-edge row tiles
//     (Z^T = W_s X^T + c[v] on the matrix cores, |z| half of the leaky ReLU as one |.|-modifier FMA per channel, permlane
//     head reduction, log2-domain online softmax, input-space aggregation, lane <-> four channels epilogue).  With a
//     hand-out order (sorted by decreasing degree) the phase ends at the first isolated destination - 94 % of the agents
//     of a random-policy rollout never enter it.
//
//  N  the `near` relation of ALL destinations + the residual-only `seen` half of the isolated ones, on BLOCKS OF 16
//     DESTINATIONS.  MFMA column j = destination j of the block; one score tile per EDGE SLOT u (the u-th in-edge of every
//     destination; the destination term rides in the contraction: A = [W_s | W_d] rows, B = (x_u0, x_u1, x_v0, x_v1)), so
//       * lane (j, k) ends up with the scores of ALL in-edges of destination j for head k in registers: the segment
//         softmax and the input-space aggregation are lane-local (no cross-lane step, no padding: n - 1 = 7 neighbours
//         are 7 tiles, not 8 columns of 16), for 16 destinations x 4 heads at once;
//       * destination meta data is a plain vector load (lane j <-> destination j): no hand-over through v_readlane;
//       * the per-destination epilogue - ReLU(W_s agg_k + has b_s + W_r x_v + b_r) for 256 channels, and ReLU(W_r x_v +
//         b_r) of `seen` for isolated destinations - is ONE MORE MATRIX-CORE PRODUCT per 16 channels x 16 destinations:
//         K groups = (agg_k0, agg_k1, x_v0, x_v1) as exact bf16 triples, biases in the free K slots against (has, 1);
//         the lane <-> channel FMAs of rounds 1-3 (44 VALU per destination) become 2 MFMAs + 8 v_max per destination;
//       * the D layout (lane = 4 channels of ONE destination) is turned into full rows through a per-wave LDS buffer:
//         16 destinations x 512 B per pass, written with ds_write_b128, read back row-contiguously and stored as 512-B
//         segments (two destinations per store instruction).
//
// All bf16 A operands (score tiles of both phases, both epilogue products: 4 x 16 KB) are built ONCE PER WORKGROUP into
// LDS by all 512 threads (rounds 1-3: every wavefront split its own, ~650 VALU per wave - a quarter of a rollout launch's
// instructions); the per-wave prologue of a phase is 16 ds_read_b128 + 16 for the attention vector.
//
// Arithmetic: fp32 in / out / accumulate.  Score GEMM and epilogue product run on the bf16 matrix cores as six exact bf16
// products per fp32 product (bf16x3.h: every product exact in the fp32 accumulator, the three dropped ones <= 2^-23 of
// the product): results equal the fp32-MFMA build's to fp32 rounding, not bit for bit.
// Instantiated for D = 64 (H = 256: every BASELINE configuration); other shapes use the per-relation kernels.
#include "common.h"
#include "k1_x3.h"

namespace uavgnn {

int gatv2_hetero_launch_f32(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order, const float* x_ubs,
                            const int32_t* near_off, const float* x_dst, int N, const float* const* seen_params,
                            const float* const* near_params, float slope, float* out, int ld_out, float* attn_save_seen,
                            float* attn_save_near, int phases, hipStream_t st);

namespace {

#ifndef K1_STORE_FLAGS
#define K1_STORE_FLAGS " nt"   // the rows are written once and read by the next kernel from L2 / HBM: non-temporal (23.7 -> 22.4 us)
#endif
#ifndef K1_ABLATE
#define K1_ABLATE 0   // 1 (tools/ubench/k1_env_bench.hip only): `phases` bit 5 skips the score tiles of phase N, bit 6 its row stores, bit 7 its epilogue products
#endif
constexpr int kWavesPerBlock = 8;
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int NH = 4;
constexpr int D = 64;
constexpr int H = NH * D;          // 256
constexpr int CT = H / 16;         // 16 channel tiles
constexpr int TPH = D / 16;        // channel tiles per head
constexpr int FS_S = 4, FS_N = 2;
constexpr int UB = 8;              // edge slots per pass of phase N
constexpr int kBounceLd = 132;     // floats per destination row of the per-wave row buffer (128 + 4: 8 consecutive lanes of a ds_write_b128 on distinct 16-B slots)
constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int kRowRor = 0x120;        // row_ror:n

__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<kRowRor + 8>(v);
  v += dpp_mov<kRowRor + 4>(v);
  v += dpp_mov<kRowRor + 2>(v);
  v += dpp_mov<kRowRor + 1>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<kRowRor + 8>(v));
  v = fmaxf(v, dpp_mov<kRowRor + 4>(v));
  v = fmaxf(v, dpp_mov<kRowRor + 2>(v));
  v = fmaxf(v, dpp_mov<kRowRor + 1>(v));
  return v;
}
__device__ __forceinline__ int row16_max_i(int v) {
  v = max(v, dpp_mov_i<kRowRor + 8>(v));
  v = max(v, dpp_mov_i<kRowRor + 4>(v));
  v = max(v, dpp_mov_i<kRowRor + 2>(v));
  v = max(v, dpp_mov_i<kRowRor + 1>(v));
  return v;
}

// Sum pe[k] over the four 16-lane groups and leave head (lane>>4)'s total in every lane: 3 swaps + 3 adds.
__device__ __forceinline__ float reduce_heads(float pe0, float pe1, float pe2, float pe3) {
  auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe0), __float_as_uint(pe2), false, false);
  const float a = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);
  auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe1), __float_as_uint(pe3), false, false);
  const float b = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// 16-byte row-piece store under a wave-uniform 64-bit lane mask: EXEC is narrowed by scalar instructions around ONE
// global_store (address = wave-uniform base in an SGPR pair + per-lane byte offset in one VGPR).  No branch (hipcc puts an
// s_cbranch_execz around every `if (lane predicate) store`), no address VGPR pair per store (eight of them spill).
//   * s_and_b64 writes SCC: declared, or the compiler keeps a compare result live across the statement (found the hard way);
//   * no hazard recognizer runs inside inline asm: the s_nop covers "SALU writes SGPR -> VMEM reads it" (5 wait states);
//   * the compiler does not know these are stores: every s_waitcnt vmcnt(n) it places later waits for them as well, so loads
//     that are in flight are consumed BEFORE the stores of a block are issued (K1_TOUCH).
// Measured alternatives: raw buffer stores with out-of-range offsets for the masked lanes (compiler-visible, exact counts,
// fastest: 810 vs 870 us on the time-batched launch) returned WRONG DATA under store back-pressure on this hardware - the first
// dword of the last four lanes of every 16-lane group of one store of a burst of eight arrived as the NEXT store's address
// operand, in ~0.003 % of the rows, timing dependent (tools/k1_check.py finds it in seconds); plain predicated C++ stores
// (per-lane predicate, address pairs) spill and run 10 % slower.
__device__ __forceinline__ void store_piece(float* base, unsigned lane_off, f32x4 v, unsigned long long lanes) {
  asm volatile(
      "s_mov_b64 vcc, exec\n\t"
      "s_and_b64 exec, exec, %3\n\t"
      "s_nop 4\n\t"
      "global_store_dwordx4 %0, %1, %2" K1_STORE_FLAGS "\n\t"
      "s_mov_b64 exec, vcc"
      :
      : "v"(lane_off), "v"(v), "s"(base), "s"(lanes)
      : "vcc", "scc");
}
// ... and the same store for a block whose 16 destinations are all written (every block but the last of a ragged N): no EXEC
// traffic, no mask arithmetic - 9 scalar instructions per store less
__device__ __forceinline__ void store_piece_all(float* base, unsigned lane_off, f32x4 v) {
  asm volatile("s_nop 4\n\t"
               "global_store_dwordx4 %0, %1, %2" K1_STORE_FLAGS
               :
               : "v"(lane_off), "v"(v), "s"(base));
}
#define K1_TOUCH(x) asm volatile("" ::"v"(x))


__device__ __forceinline__ float rl(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

struct RelParams {   // one GATv2Conv: fc_src, fc_dst, attn, res_fc (DGL layout, gnn_agents.py:93-96)
  const float* W_s; const float* b_s; const float* W_d; const float* b_d; const float* attn; const float* W_r;
  const float* b_r;
};

// acc += |z| a as ONE v_fma_f32 with the |.| source modifier.  Left alone, hipcc's SLP vectorizer pairs the accumulators into
// v_pk_fma_f32 - which has no |.| modifier - and spends a v_and_b32 per element on the absolute value (82 instead of 64 VALU
// instructions per row tile); the empty asm makes the accumulator opaque to it.  (The FMA itself must stay a compiler-visible
// instruction: written as inline asm it escapes the hazard recognizer and reads matrix-core results too early - measured as
// 3e-5 relative errors.)
__device__ __forceinline__ void k1_fma_abs(float& acc, float z, float a) {
  acc = fmaf(a, fabsf(z), acc);
  asm("" : "+v"(acc));
}

// The MFMA + |z| FMA block shared by both phases: 16 channel tiles of one row tile -> log2-domain score of
// (column j, head g) in every lane, up to a per-(destination, head) constant that cancels in the softmax.
// The instruction order is PINNED (sched_barrier): MFMA ct+1 is issued, then the four |z| FMAs of tile ct run in its
// shadow.  Left to itself hipcc hoists / sinks the FMAs around the MFMAs and the tile takes ~10 % longer
// (tools/ubench/tile_sched.hip).
#define UAVGNN_TILE_SCORE(WA, ATT, CINIT, WLIN, XB, XBOP, E_OUT)                                        \
  {                                                                                                     \
    float pe[NH][2];                                                                                    \
    _Pragma("unroll") for (int k = 0; k < NH; ++k) {                                                    \
      pe[k][0] = WLIN[k] * (XB);                                                                        \
      pe[k][1] = 0.f;                                                                                   \
    }                                                                                                   \
    const k1_bf16x8 xb_op = XBOP;                                                                       \
    f32x4 z_cur = K1_MFMA(WA[0], xb_op, CINIT(0));                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                 \
      f32x4 z_nxt = z_cur;                                                                              \
      if (ct + 1 < CT) {                                                                                \
        z_nxt = K1_MFMA(WA[ct + 1], xb_op, CINIT(ct + 1));                                              \
        __builtin_amdgcn_sched_barrier(0);                                                              \
      }                                                                                                 \
      const int k = ct / TPH;                                                                           \
      k1_fma_abs(pe[k][0], z_cur[0], ATT[ct][0]);                                                       \
      k1_fma_abs(pe[k][1], z_cur[1], ATT[ct][1]);                                                       \
      k1_fma_abs(pe[k][0], z_cur[2], ATT[ct][2]);                                                       \
      k1_fma_abs(pe[k][1], z_cur[3], ATT[ct][3]);                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      z_cur = z_nxt;                                                                                    \
    }                                                                                                   \
    E_OUT = reduce_heads(pe[0][0] + pe[0][1], pe[1][0] + pe[1][1], pe[2][0] + pe[2][1], pe[3][0] + pe[3][1]); \
  }

enum { kSetSeen = 0, kSetNear = 1, kSetEpiNear = 2, kSets = 3 };
enum { K1_TILES = 0, K1_NEAR = 1, K1_SEEN = 2 };
#ifndef K1_ORDER
#define K1_ORDER 0   // (only the ablation harness selects an order by hand: `phases` bits 12-13 = order + 1)
#endif
// order 0: wave A (w < 4) seen, tiles, near / wave B (its SIMD mate w + 4) tiles, seen, near; 1: both seen, tiles, near;
// 2: A tiles, near, seen / B seen, tiles, near
__device__ __forceinline__ int k1_part(int order, bool tiles_first, int step) {
  if (order == 1) return step == 0 ? K1_SEEN : step == 1 ? K1_TILES : K1_NEAR;
  if (order == 2) return tiles_first ? (step == 0 ? K1_TILES : step == 1 ? K1_NEAR : K1_SEEN)
                                     : (step == 0 ? K1_SEEN : step == 1 ? K1_TILES : K1_NEAR);
  return tiles_first ? (step == 0 ? K1_TILES : step == 1 ? K1_SEEN : K1_NEAR) : (step == 0 ? K1_SEEN : step == 1 ? K1_TILES : K1_NEAR);
}

// What a workgroup needs from the parameters of the two GATv2Conv modules, in the layout its LDS holds it: the bf16 A operands of
// the score tiles of both phases and of the `near` epilogue product ([set][channel tile][lane]), the fp32 operands of phase S's
// lane <-> channel prologue / epilogue, the attention vectors x log2(e) (1 - slope) / 2 and wa[rel][k][f] = log2(e) (1 + slope) /
// 2 x sum_d attn[k,d] W_s[k,d,f].  Built by every workgroup in its prologue (k1_build_image) - or ONCE by
// uavgnn_gatv2_hetero_prepare into a caller-provided buffer that the workgroups only copy (one round trip instead of one round
// trip + ~200 VALU per wavefront: 2.9 -> 1.x us of a 20-us rollout launch; worth it when the same weights serve many launches).
struct K1Image {
  k1_u32x4 A[kSets][CT * kWave];
  float Ws[H * FS_S];                  // fc_src.weight of `seen`, row-major [H, 4]
  float Wds[H * 2], Wrs[H * 2];        // seen fc_dst / res_fc
  float Bss[H], Bds[H], Brs[H];        // seen biases (b_r: 0 when absent)
  float As[H], An[H];
  float Wa[2][NH * 4];
};
static_assert(sizeof(K1Image) % 16 == 0, "copied in 16-byte pieces");

// The image is built in two halves so that a caller can put its own dependent loads between them: k1_load_raw issues every
// global load (branch-free: ONE round trip; a version with `if (tid < ...)` blocks and null-pointer branches around the loads
// cost ten dependent round trips = 4 us of a 20-us launch), k1_store_image converts and writes LDS with unconditional stores.
struct K1Raw {
  f32x4 ws, wd, wr;
  float as, an, bs, bd, br;
  float wa_a[4], wa_w[4];
  float w[2 * kSets], b[2 * kSets];   // 3 sets x 1024 entries over 512 threads: weight and bias of entry q * 512 + tid
};

__device__ __forceinline__ K1Raw k1_load_raw(const RelParams& ps, const RelParams& pn, int tid) {
  K1Raw r;
  const float* const brs_p = ps.b_r != nullptr ? ps.b_r : ps.b_s;   // res_fc.bias may be absent: any valid address, scaled by 0
  const float* const brn_p = pn.b_r != nullptr ? pn.b_r : pn.b_s;
  const float brn_on = pn.b_r != nullptr ? 1.f : 0.f;
  const int t256 = tid & 255, t128 = tid & 127;
  r.ws = reinterpret_cast<const f32x4*>(ps.W_s)[t256];
  r.wd = reinterpret_cast<const f32x4*>(ps.W_d)[t128];
  r.wr = reinterpret_cast<const f32x4*>(ps.W_r)[t128];
  r.as = ps.attn[t256];
  r.an = pn.attn[t256];
  r.bs = ps.b_s[t256];
  r.bd = ps.b_d[t256];
  r.br = brs_p[t256];
  // wa[rel][k][f]: 2 x 16 outputs, 16 partial sums of 4 terms each; 512 threads = 32 rows of 16 lanes
  const int wa_kf = tid >> 4, wa_part = tid & 15;
  const int wa_rel = wa_kf >> 4, wa_k = (wa_kf >> 2) & 3, wa_f = wa_kf & 3;
  {
    const float* at = wa_rel ? pn.attn : ps.attn;
    const float* ws = wa_rel ? pn.W_s : ps.W_s;
    const int F = wa_rel ? FS_N : FS_S, f = min(wa_f, F - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = wa_k * D + wa_part + 16 * i;
      r.wa_a[i] = at[d];
      r.wa_w[i] = ws[d * F + f];
    }
  }
#pragma unroll
  for (int q = 0; q < 2 * kSets; ++q) {
    const int idx = (q & 1) * kThreads + tid;
    const int row = (idx >> 6) * 16 + (idx & 15), gg = (idx >> 4) & 3;
    if ((q >> 1) == kSetSeen) {
      r.w[q] = ps.W_s[row * FS_S + gg];
      r.b[q] = 0.f;
    } else if ((q >> 1) == kSetNear) {   // A = [W_s | W_d], the channel bias b_s + b_d in the free slots of K groups 0 / 1
      r.w[q] = *(gg < 2 ? pn.W_s + row * FS_N + gg : pn.W_d + row * 2 + gg - 2);
      r.b[q] = pn.b_s[row] + pn.b_d[row];
    } else {   // epilogue of `near`: A = [W_s | W_r]; b_s against `has` in K groups 0 / 1, b_r against 1 in groups 2 / 3
      r.w[q] = *(gg < 2 ? pn.W_s + row * FS_N + gg : pn.W_r + row * 2 + gg - 2);
      r.b[q] = *(gg < 2 ? pn.b_s + row : brn_p + row) * (gg < 2 ? 1.f : brn_on);
    }
  }
  return r;
}

// unconditional stores (two / four threads write the same value to the same word): no branch, no pessimistic wait
__device__ __forceinline__ void k1_store_image(K1Image& sI, const K1Raw& r, bool seen_res_bias, float slope, int tid) {
  const float c_abs = kLog2e * 0.5f * (1.f - slope), c_lin = kLog2e * 0.5f * (1.f + slope);
  const int t256 = tid & 255, t128 = tid & 127;
  reinterpret_cast<f32x4*>(sI.Ws)[t256] = r.ws;
  reinterpret_cast<f32x4*>(sI.Wds)[t128] = r.wd;
  reinterpret_cast<f32x4*>(sI.Wrs)[t128] = r.wr;
  sI.As[t256] = c_abs * r.as;
  sI.An[t256] = c_abs * r.an;
  sI.Bss[t256] = r.bs;
  sI.Bds[t256] = r.bd;
  sI.Brs[t256] = seen_res_bias ? r.br : 0.f;
  {
    const int wa_kf = tid >> 4, wa_part = tid & 15;
    const int wa_rel = wa_kf >> 4, wa_f = wa_kf & 3;
    float a0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) a0 = fmaf(r.wa_a[i], r.wa_w[i], a0);
    a0 = row16_sum(a0);
    if (wa_part == 0) sI.Wa[wa_rel][wa_kf & 15] = (wa_f < (wa_rel ? FS_N : FS_S)) ? c_lin * a0 : 0.f;
  }
#pragma unroll
  for (int q = 0; q < 2 * kSets; ++q) {
    const int idx = (q & 1) * kThreads + tid;
    const int gg = (idx >> 4) & 3;
    const int set = q >> 1;
    const int part = set == kSetSeen ? 0 : set == kSetNear ? (gg == 0 ? 1 : gg == 1 ? 2 : 0) : ((gg & 1) ? 2 : 1);
    sI.A[set][idx] = k1_a_operand(r.w[q], r.b[q], part);
  }
}

// One workgroup builds the image in LDS exactly as the forward's prologue does and copies it out.
__global__ __launch_bounds__(kThreads) void gatv2_hetero_prepare_kernel(RelParams ps, RelParams pn, float slope,
                                                                        k1_u32x4* __restrict__ image) {
  __shared__ K1Image sI;
  const K1Raw raw = k1_load_raw(ps, pn, threadIdx.x);
  k1_store_image(sI, raw, ps.b_r != nullptr, slope, threadIdx.x);
  __syncthreads();
  const k1_u32x4* src = reinterpret_cast<const k1_u32x4*>(&sI);
  for (int i = threadIdx.x; i < static_cast<int>(sizeof(K1Image) / 16); i += kThreads) image[i] = src[i];
}

// SAVE: the attention weights of both relations are written for the backward pass (training forwards); the inference
// instantiation does not carry them through the block.
// IMG: the parameter image is copied from `image` (uavgnn_gatv2_hetero_prepare) instead of built (a template parameter, not a
// branch: behind a run-time branch the compiler merges the load counts of the two sides pessimistically and the prologue waits
// for its first round trip before it issues the second).
// RM: the maxima of the `near` half and of the `seen` half of every output row are written to rm_near / rm_seen [N] - the two row bounds
// of the f16x2 f_aggr product behind a TIME-BATCHED launch (csrc/gemm_h2.hip takes the larger).  Every row's halves pass through registers
// here and each element of the two arrays has exactly one writer (phase N: the near half of every destination and the residual-only seen
// half of the isolated ones; phase S: the seen half of the others); rows are >= 0 behind the ReLU.  A template parameter: the plain
// instantiations keep their register allocation (the rollout launch is store-bound and pays +1.9 us for the maxima: not used there).
template <bool SAVE, bool IMG, bool RM = false>
__global__ __launch_bounds__(kThreads, 2) void gatv2_hetero_fwd_kernel(
    const float* __restrict__ x_gt, const int32_t* __restrict__ seen_off, const int32_t* __restrict__ seen_order,
    const float* __restrict__ x_ubs, const int32_t* __restrict__ near_off, const float* __restrict__ x_dst, int N,
    int E_seen, RelParams ps, RelParams pn, float slope, float* __restrict__ out, int ld_out, float* __restrict__ a_save_s_arg,
    float* __restrict__ a_save_n_arg, int phases, const k1_u32x4* __restrict__ image, float* __restrict__ rm_near,
    float* __restrict__ rm_seen) {
  constexpr bool rm_on = RM;
  float* const a_save_n = SAVE ? a_save_n_arg : nullptr;
  float* a_save_s = SAVE ? a_save_s_arg : nullptr;   // (the host launches the SAVE instantiation when either buffer is given)
  // everything a workgroup needs from the parameters (K1Image, 61 KB): built here, or copied from a caller-provided image
  __shared__ K1Image sI;
  auto& sA = sI.A;
  auto& sWs = sI.Ws;
  auto& sWds = sI.Wds;
  auto& sWrs = sI.Wrs;
  auto& sBss = sI.Bss;
  auto& sBds = sI.Bds;
  auto& sBrs = sI.Brs;
  auto& sAs = sI.As;
  auto& sAn = sI.An;
  auto& sWa = sI.Wa;
  __shared__ __attribute__((aligned(16))) float sC[kWavesPerBlock][H];   // phase S: destination term; phase N: aggregate hand-over
  __shared__ __attribute__((aligned(16))) float sRow[kWavesPerBlock][16 * kBounceLd];   // phase N: output rows of one pass
  __shared__ int sS[kWavesPerBlock][3 * kWave];   // first hand-out chunk of phase S, requested in the prologue (+ the upper half of sC)

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15;          // MFMA column (edge slot / destination of the block) / A row
  const int g = lane >> 4;          // lane group: K group of the A / B operands, head after the reduction

  const int stride = gridDim.x * kWavesPerBlock;
  const int it0 = blockIdx.x * kWavesPerBlock + wave;
#ifndef K1_S_TRANSPOSE
#define K1_S_TRANSPOSE 1
#endif
  // Phase S hands the degree-SORTED destinations out by position: position p, p + stride, ... go to one wavefront.  With p = it0 the
  // eight heaviest destinations of a rollout batch (ONE destination per wavefront there: 1 966 of 32 768 have in-edges at D-env) all
  // land on workgroup 0 - four row tiles each, on both wavefronts of every SIMD - and the last workgroups get none.  Transposed
  // (wavefront w of workgroup b starts at position w * gridDim + b) every CU gets one destination of each weight class and the two
  // wavefronts of a SIMD (w, w + 4) a heavy and a light one: D-env rollout launch 19.05 -> 18.15 us, same box, same rows bit for bit
  // (profiles/r06_k1_handout_ab.txt).  Only for rollout-size launches (one block of phase N per wavefront): over many chunks of
  // `stride` positions the transposed order gives wavefront 0 the heaviest eighth of EVERY chunk (time-batched launch 751 -> 786 us),
  // while the plain order's skew between workgroups shrinks with the number of chunks.
  const bool one_chunk = ((N + 15) >> 4) <= stride;
  const int it0s = (K1_S_TRANSPOSE && one_chunk) ? wave * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x) : it0;
  const int nblk = (N + 15) >> 4;   // blocks of 16 destinations (phase N)
#if K1_ABLATE   // `phases` bit 10: per-wavefront time stamps (100 MHz s_memrealtime) into the buffer passed as attn_save_seen
  unsigned long long* const dbg = (phases & 1024) ? reinterpret_cast<unsigned long long*>(a_save_s_arg) : nullptr;
  if (phases & 1024) a_save_s = nullptr;
#define K1_STAMP(i) if (dbg != nullptr && lane == 0) dbg[static_cast<size_t>(it0) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define K1_STAMP(i)
#endif
  K1_STAMP(0)

  // Inputs of this wavefront's FIRST block of phase N are requested before the workgroup prologue (their two dependent
  // round trips overlap the staging below): destination meta data - lane j <-> destination j, a plain vector load -, then
  // the first-pass edge features.  Clamped, never predicated (see phase S).
  struct Meta { int n0, n1, s0, s1; float2 xv; };
  auto load_meta = [&](const int blk) {
    const int vc = min(blk * 16 + j, N - 1);
    Meta m;
    m.n0 = near_off[vc];
    m.n1 = near_off[vc + 1];
    m.s0 = seen_off[vc];
    m.s1 = seen_off[vc + 1];
    m.xv = *reinterpret_cast<const float2*>(x_dst + 2 * vc);
    return m;
  };
  float2 xq[UB];   // first-pass inputs of the next block: slot u of destination j (masked slots read edge row 0)
  auto load_edges = [&](const Meta& m, const int blk, const int base) {
    const int dg = (blk * 16 + j < N) ? m.n1 - m.n0 : 0;
#pragma unroll
    for (int u = 0; u < UB; ++u)
      xq[u] = *reinterpret_cast<const float2*>(x_ubs + static_cast<size_t>(base + u < dg ? m.n0 + base + u : 0) * FS_N);
  };
  Meta mnext = load_meta(min(it0, nblk - 1));
  // ... and so is the first hand-out chunk of phase S (destination, segment, features of 64 positions: two of the three
  // dependent round trips of a phase that is pure latency on a rollout batch); it waits in LDS while phase N runs
  const int s_it = min(it0s + lane * stride, N - 1);
  const int p_v = seen_order != nullptr ? seen_order[s_it] : s_it;

  // ---- workgroup prologue: the parameter image into LDS ------------------------------------------------------------------
  __builtin_amdgcn_sched_barrier(0);   // order pinned: meta data (head of the dependent chain), parameters, dependent loads
  // First half: every load that does not depend on another one.  With a prepared image (uavgnn_gatv2_hetero_prepare) that is a
  // copy, 8 x 16 bytes per thread; else the parameters themselves.
  constexpr int kPieces = sizeof(K1Image) / 16, kPer = (kPieces + kThreads - 1) / kThreads;
  k1_u32x4 piece[kPer];
  K1Raw raw;
  if constexpr (IMG) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) piece[i] = image[min(tid + i * kThreads, kPieces - 1)];
  } else {
    raw = k1_load_raw(ps, pn, tid);
  }
  __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks the parameter loads below the wait for the meta data)
  // the one dependent round trip: needs the meta data requested first
  load_edges(mnext, min(it0, nblk - 1), 0);
  const int p_e0 = seen_off[p_v], p_e1 = seen_off[p_v + 1];
  const float2 p_xv = *reinterpret_cast<const float2*>(x_dst + 2 * p_v);
  // second half: the image into LDS
  if constexpr (IMG) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) reinterpret_cast<k1_u32x4*>(&sI)[min(tid + i * kThreads, kPieces - 1)] = piece[i];
  } else {
    k1_store_image(sI, raw, ps.b_r != nullptr, slope, tid);
  }
  // Workgroup barrier WITHOUT the memory fence of __syncthreads() (s_waitcnt vmcnt(0)): only the LDS image has to be complete.
  // The second round trip of the prologue - first-pass edges of phase N, segment bounds of phase S - stays in flight behind the
  // barrier, and the residual-only `seen` rows (which need the destination meta data only) are on their way one round trip
  // earlier.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  K1_STAMP(1)
  // per-wave hand-over of phase S's first chunk (read back by this wavefront only), stored where the second round trip is
  // waited for anyway: in front of the first score tiles, or in front of phase S when the wavefront has no block
  auto stash = [&] {
    int* st = reinterpret_cast<int*>(sC[wave]) + 2 * kWave;
    st[lane] = p_v;
    st[kWave + lane] = p_e0;
    sS[wave][lane] = p_e1;
    sS[wave][kWave + lane] = __float_as_int(p_xv.x);
    sS[wave][2 * kWave + lane] = __float_as_int(p_xv.y);
  };
  const bool phase_n = (phases & 2) && it0 < nblk;
  if (!phase_n) stash();

  const bool phase_s = it0s < N && (phases & 1) && E_seen > 0;
  // (Requesting phase S's first tile from inside phase N, in front of the `near` row stores of the wavefront's last block, was
  // measured: +0.5 us on a rollout launch - the loads queue behind 33 MB of stores and the wait for them is no shorter.  So was
  // running the phases in opposite order on the two wavefronts of a SIMD: +2.5 us - phase S's loads crawl while the other
  // wavefronts' rows fill the store queues, and the registers kept across the phase loop spill.)

  float* __restrict__ cw = sC[wave];
  const f32x4 czero = {0.f, 0.f, 0.f, 0.f};

  // ====== phase N: `near` + residual-only `seen` rows on blocks of 16 destinations (column j <-> destination) ========
  // Runs BEFORE phase S: it writes 2 KB per destination, and its row stores drain while phase S computes.
  if (phase_n) {
    const unsigned one_tile = g == 0 ? 0x3F803F80u : g == 1 ? 0x00003F80u : 0u;   // bf16 1.0 against the bias slots of the A operand
    float* __restrict__ rw = sRow[wave];
    const int half = lane >> 5, c16 = lane & 31;   // row stores: destination 2i + half of the block, 16-byte chunk c16 of the pass
    const unsigned row_off = static_cast<unsigned>(half * ld_out + 4 * c16) * 4u;   // byte offset of the lane's piece from the row pair's base
#if K1_ABLATE
    if ((phases & 512) && wave < 4) __builtin_amdgcn_s_setprio(2);
#endif

    K1_STAMP(2)
    // The two wavefronts that share a SIMD (waves w and w + 4 of the workgroup) run the store-bound part (residual-only `seen`
    // rows: LDS + memory pipeline) and the issue-bound part (score tiles: VALU + matrix cores) of a block in OPPOSITE order,
    // so that one of them always has instructions to issue (in lock step - both storing, then both computing - the phase took
    // 11.8 us of a rollout launch for 5.5 us of issue time).
    const bool tiles_first = (wave & 4) != 0;
    // one block per wavefront (a rollout launch): both wavefronts of a SIMD emit the residual-only rows first - the earlier the
    // 31 MB of them are on their way, the shorter the HBM-bound tail (measured 19.2 vs 20.8 us); several blocks per wavefront (the
    // time-batched launches): opposite order on the two wavefronts of a SIMD (830 vs 945 us)
    const int order = (K1_ABLATE && (phases & (3 << 12))) ? ((phases >> 12) & 3) - 1 : (nblk > stride ? 0 : 1);
    for (int blk = it0; blk < nblk; blk += stride) {
      const Meta mt = mnext;
      const int v0 = blk << 4;
      const bool valid_v = v0 + j < N;
      const int deg = valid_v ? mt.n1 - mt.n0 : 0;
      const bool iso = valid_v && mt.s1 == mt.s0;
      float2 (&xu)[UB] = xq;
      const int nb = min(blk + stride, nblk - 1);
      mnext = load_meta(nb);   // meta data of the next block: in flight while this one computes
      const int maxdeg = __builtin_amdgcn_readfirstlane(row16_max_i(deg));
      const float xvg = (g & 1) ? mt.xv.y : mt.xv.x;   // only read by lane groups 2, 3
      const unsigned vmask = static_cast<unsigned>(__builtin_amdgcn_ballot_w64(valid_v)) & 0xffffu;
      const unsigned imask = static_cast<unsigned>(__builtin_amdgcn_ballot_w64(iso)) & 0xffffu;
#if K1_ABLATE   // `phases` bit 11: every block stores to the first rows (no HBM drain: what the store INSTRUCTIONS cost)
      float* const row0 = out + static_cast<size_t>((phases & 2048) ? (blockIdx.x & 7) * 16 : v0) * ld_out;
#else
      float* const row0 = out + static_cast<size_t>(v0) * ld_out;   // wave-uniform
#endif

      // ---- epilogue products + row stores: 8 channel tiles (512 B per destination) per pass ------------------------
      // D[channel][destination]: lane (j, g) holds channels 16 ct + 4 g .. + 3 of destination j = one 16-byte piece of its
      // row; the pieces go through the per-wave LDS buffer and leave as 512-byte row segments, two destinations per
      // store instruction (lane half <-> destination: the predicate of a store is a 64-bit EXEC mask built by scalar
      // instructions from the 16-bit destination mask; no branch, so the number of stores in flight is static and the
      // prefetches behind them are waited for with counted s_waitcnt).  Storing from the transposed product D[destination]
      // [channel] without LDS - 64-byte pieces per 16 lanes - was measured: four times the store instructions, 12 cycles
      // each in the memory pipeline, slower.
      float nmx = 0.f;   // RM: running maximum of this lane's pieces of the `near` rows (lane (j, g): destination j)
      auto emit = [&](const int set, const int col0, const unsigned mask, const k1_bf16x8* bop, const bool per_head) {
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int ct = hp * 8 + c;
            const k1_u32x4 a = sA[set][ct * kWave + lane];
            f32x4 d = K1_MFMA(a, bop[per_head ? ct / TPH : 0], czero);
#if K1_ABLATE
            if (phases & 128) d = czero;
#endif
            d[0] = fmaxf(d[0], 0.f);
            d[1] = fmaxf(d[1], 0.f);
            d[2] = fmaxf(d[2], 0.f);
            d[3] = fmaxf(d[3], 0.f);
            if (rm_on) nmx = fmaxf(fmaxf(nmx, d[0]), fmaxf(fmaxf(d[1], d[2]), d[3]));
            *reinterpret_cast<f32x4*>(rw + j * kBounceLd + c * 16 + 4 * g) = d;
          }
          wave_sync_lds();
#if K1_ABLATE
          if (!(phases & 64))
#endif
          {
            f32x4 vals[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vals[i] = *reinterpret_cast<const f32x4*>(rw + (2 * i + half) * kBounceLd + 4 * c16);
            if (mask == 0xffffu) {   // wave-uniform
#pragma unroll
              for (int i = 0; i < 8; ++i) store_piece_all(row0 + static_cast<size_t>(2 * i) * ld_out + col0 + hp * 128, row_off, vals[i]);
            } else {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const unsigned long long m64 = (((mask >> (2 * i)) & 1u) ? 0x00000000ffffffffull : 0ull) |
                                               (((mask >> (2 * i + 1)) & 1u) ? 0xffffffff00000000ull : 0ull);
                store_piece(row0 + static_cast<size_t>(2 * i) * ld_out + col0 + hp * 128, row_off, vals[i], m64);
              }
            }
          }
          wave_sync_lds();   // the row buffer is rewritten by the next pass
        }
      };

      float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f;
      float pw[UB];
      // Order of the three parts of a block - K1_SEEN: residual-only `seen` rows (store-bound), K1_TILES: score tiles + softmax
      // (issue-bound), K1_NEAR: `near` rows (store-bound, needs the tiles).
      for (int step = 0; step < 3; ++step) {
        const int what = k1_part(order, tiles_first, step);
        if (what == K1_SEEN) {
          // The residual-only `seen` rows of the isolated destinations, ReLU(W_r x_v + b_r): two FMAs per channel, no
          // contraction worth a matrix-core product - lane <-> four consecutive channels, one full 1-KB row per store
          // instruction straight from the registers (as a matrix-core product + LDS transposition like the `near` rows the
          // part took 2.2-3.9 us of a rollout launch, bound by the 16-byte LDS writes and the store pipeline).  Issued in one
          // burst the 16 stores hold the wavefront for ~2.8 us on a full store queue; spread over the score tiles, two per
          // tile, they hold the TILES up instead (measured: 22.2 vs 20.0 us) - the burst stays.
          if (imask != 0u) {
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));   // constants re-read per block, not held across the tile part
            const float4 br4 = reinterpret_cast<const float4*>(sBrs)[lane_s];
            const float4 wr_lo = reinterpret_cast<const float4*>(sWrs)[2 * lane_s], wr_hi = reinterpret_cast<const float4*>(sWrs)[2 * lane_s + 1];
            float* const rs = row0 + 4 * lane;
            unsigned su[16];    // RM: maximum of the residual-only row of destination d (scalar registers)
#pragma unroll
            for (int d = 0; d < 16; ++d) su[d] = 0u;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
              if ((imask >> d) & 1u) {   // wave-uniform
                const float x0 = rl(mt.xv.x, d), x1 = rl(mt.xv.y, d);
                f32x4 o;
                o[0] = fmaxf(fmaf(wr_lo.y, x1, fmaf(wr_lo.x, x0, br4.x)), 0.f);
                o[1] = fmaxf(fmaf(wr_lo.w, x1, fmaf(wr_lo.z, x0, br4.y)), 0.f);
                o[2] = fmaxf(fmaf(wr_hi.y, x1, fmaf(wr_hi.x, x0, br4.z)), 0.f);
                o[3] = fmaxf(fmaf(wr_hi.w, x1, fmaf(wr_hi.z, x0, br4.w)), 0.f);
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(rs + static_cast<size_t>(d) * ld_out));
                if (rm_on) {   // 16-lane rows by DPP, the four rows on the scalar unit (values >= 0: integer max of the bit patterns)
                  const unsigned mu = __float_as_uint(row16_max(fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3]))));
                  su[d] = max(max(__builtin_amdgcn_readlane(mu, 0), __builtin_amdgcn_readlane(mu, 16)),
                              max(__builtin_amdgcn_readlane(mu, 32), __builtin_amdgcn_readlane(mu, 48)));
                }
              }
            }
            if (rm_on) {
              unsigned sres = 0u;
#pragma unroll
              for (int d = 0; d < 16; ++d) sres = (lane == d) ? su[d] : sres;
              if (lane < 16 && ((imask >> lane) & 1u)) rm_seen[v0 + lane] = __uint_as_float(sres);
            }
          }
          if (blk == it0) { K1_STAMP(3) }
        } else if (what == K1_TILES) {
          // A operands and attention vector of the score tiles: (re)read from LDS per block - 32 ds_read_b128 - instead of held
          // across the row-store parts of the block, whose LDS round trips want the registers (the opaque lane index keeps the
          // compiler from hoisting the reads out of the block loop)
#pragma unroll
          for (int u = 0; u < UB; ++u) K1_TOUCH(xu[u].x);   // in flight since the previous block (or the prologue): landed before the row stores of `near`
          if (blk == it0) stash();
          int lane_v = lane;
          asm volatile("" : "+v"(lane_v));
          k1_u32x4 Wa[CT];
          float att[CT][4], wlin[NH];
  #pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            Wa[ct] = sA[kSetNear][ct * kWave + lane_v];
            const float4 a4 = *reinterpret_cast<const float4*>(sAn + ct * 16 + 4 * (lane_v >> 4));
            att[ct][0] = a4.x; att[ct][1] = a4.y; att[ct][2] = a4.z; att[ct][3] = a4.w;
          }
  #pragma unroll
          for (int k = 0; k < NH; ++k) wlin[k] = (g < 2) ? sWa[1][k * 4 + (lane_v >> 4)] : 0.f;
          for (int base = 0; base < maxdeg; base += UB) {
            if (base > 0) {   // degrees above 8: further passes through the online softmax
              load_edges(mt, blk, base);
            }
            float e[UB];
  #pragma unroll
            for (int u = 0; u < UB; ++u) {
              e[u] = 0.f;
              if (base + u < maxdeg) {   // wave-uniform
                const float xB = (g == 0) ? xu[u].x : (g == 1) ? xu[u].y : xvg;
  #if K1_ABLATE
                if (phases & 32) e[u] = xB; else
  #endif
  #define K1_CINIT_N(ct) czero
                UAVGNN_TILE_SCORE(Wa, att, K1_CINIT_N, wlin, xB, k1_b_operand(xB, one_tile), e[u])
              }
            }
            // ---- lane-local segment softmax over the slots of this pass (lane (j, k): destination j, head k) -----------
            float mx = -INFINITY;
  #pragma unroll
            for (int u = 0; u < UB; ++u) {
              e[u] = (base + u < deg) ? e[u] : -INFINITY;
              mx = fmaxf(mx, e[u]);
            }
            const float mn = fmaxf(m, mx);
            const float mref = (mn == -INFINITY) ? 0.f : mn;      // destination without in-edges so far: every weight is exp2(-inf) = 0
            const float sc = __builtin_amdgcn_exp2f(m - mref);     // exp2(-inf) = 0 on the first pass
            den *= sc;
            s0 *= sc;
            s1 *= sc;
  #pragma unroll
            for (int u = 0; u < UB; ++u) {
              pw[u] = __builtin_amdgcn_exp2f(e[u] - mref);
              den += pw[u];
              s0 = fmaf(pw[u], xu[u].x, s0);
              s1 = fmaf(pw[u], xu[u].y, s1);
            }
            m = mn;
            if (a_save_n != nullptr && maxdeg > UB) {   // raw scores now, weights once the maximum and the sum are final
  #pragma unroll
              for (int u = 0; u < UB; ++u)
                if (base + u < deg) a_save_n[static_cast<size_t>(mt.n0 + base + u) * NH + g] = e[u];
            }
          }
          if (blk == it0) { K1_STAMP(4) }
        } else {
          const float inv = den > 0.f ? __builtin_amdgcn_rcpf(den) : 0.f;      // isolated destination: aggregate = 0
          if (a_save_n != nullptr) {
            if (maxdeg <= UB) {
    #pragma unroll
              for (int u = 0; u < UB; ++u)
                if (u < deg) a_save_n[static_cast<size_t>(mt.n0 + u) * NH + g] = pw[u] * inv;
            } else {
              for (int u = 0; u < deg; ++u) {
                float* ap = a_save_n + static_cast<size_t>(mt.n0 + u) * NH + g;
                *ap = __builtin_amdgcn_exp2f(*ap - m) * inv;
              }
            }
          }
          // ---- hand the aggregates of head k over to the lanes that feed K group g of the epilogue product ---------------
          cw[j * 8 + g] = s0 * inv;
          cw[j * 8 + 4 + g] = s1 * inv;
          wave_sync_lds();
          const f32x4 ag = *reinterpret_cast<const f32x4*>(cw + j * 8 + (g & 1) * 4);   // feature g of heads 0..3 (groups 0, 1)
          const unsigned has_w = deg > 0 ? 0xffffffffu : 0u;
          const unsigned one_epi = (g & 1 ? 0x00003F80u : 0x3F803F80u) & (g < 2 ? has_w : 0xffffffffu);
          k1_bf16x8 bop[NH];
    #pragma unroll
          for (int k = 0; k < NH; ++k) bop[k] = k1_b_operand(g < 2 ? ag[k] : xvg, one_epi);
          wave_sync_lds();   // cw is rewritten by the next block
          emit(kSetEpiNear, H, vmask, bop, true);
          if (rm_on) {
            float t = fmaxf(nmx, __shfl_xor(nmx, 16));
            t = fmaxf(t, __shfl_xor(t, 32));
            if (g == 0 && valid_v) rm_near[v0 + j] = t;
          }
          if (blk == it0) { K1_STAMP(5) }
        }
      }
      load_edges(mnext, nb, 0);   // first-pass inputs of the next block
    }
#if K1_ABLATE
    if (dbg != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K1_STAMP(6)
#endif
#if K1_ABLATE
    if (phases & 512) __builtin_amdgcn_s_setprio(0);
#endif
  }

  // =========================== phase S: `seen` on the destinations that have in-edges ===============================
  if (phase_s) {
    constexpr int kStashTiles = (16 * kBounceLd) / kWave;   // 33 row tiles = 528 in-edges fit the row buffer
    float* __restrict__ srow = sRow[wave];
    k1_u32x4 Wa[CT];
    float att[CT][4], wlin[NH];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      Wa[ct] = sA[kSetSeen][ct * kWave + lane];
      const float4 a4 = *reinterpret_cast<const float4*>(sAs + ct * 16 + 4 * g);
      att[ct][0] = a4.x; att[ct][1] = a4.y; att[ct][2] = a4.z; att[ct][3] = a4.w;
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) wlin[k] = sWa[0][k * 4 + g];
    // The per-destination constants of the prologue / epilogue (lane <-> channels 4*lane .. 4*lane+3: fc_dst, res_fc rows,
    // biases) are READ FROM LDS where they are used, once per destination: the bf16 A operands take 64 VGPRs and 28 more
    // registers held across the tile loop would spill.
    // inputs of the row tile that is processed next (possibly the first tile of the next destination)
    float4 xr;
    float xBn;
    // ALWAYS exactly two loads (clamped address, never predicated): a predicated load makes the number of outstanding
    // loads unknown to the compiler, which then drains the queue (s_waitcnt vmcnt(0)) right behind the request - the
    // prefetch would not overlap anything.  Columns j >= count read edge row 0; they are masked out of the softmax.
    // (A queue of two / three / four tiles in flight filled by an independent cursor, with the C operands re-read from LDS per
    // tile to pay for its registers, was measured: D-env rollout 20.1 vs 19.3 us, D-dense phase S 94 vs 79 us.)
    auto request = [&](const int e_first, const int count) {   // columns j < count of the tile starting at edge e_first
      const size_t u = (j < count) ? static_cast<size_t>(e_first + j) : 0;
      xr = *reinterpret_cast<const float4*>(x_gt + u * FS_S);
      xBn = x_gt[u * FS_S + g];
    };

    auto process = [&](const int v, const int ce0, const int cdeg, const float cxv0, const float cxv1, const int n_e0,
                       const int n_deg) {
      {
        const float4 bs4 = reinterpret_cast<const float4*>(sBss)[lane], bd4 = reinterpret_cast<const float4*>(sBds)[lane];
        const float4 wd_lo = reinterpret_cast<const float4*>(sWds)[2 * lane], wd_hi = reinterpret_cast<const float4*>(sWds)[2 * lane + 1];
        f32x4 cv;
        cv[0] = fmaf(wd_lo.y, cxv1, fmaf(wd_lo.x, cxv0, bs4.x + bd4.x));
        cv[1] = fmaf(wd_lo.w, cxv1, fmaf(wd_lo.z, cxv0, bs4.y + bd4.y));
        cv[2] = fmaf(wd_hi.y, cxv1, fmaf(wd_hi.x, cxv0, bs4.z + bd4.z));
        cv[3] = fmaf(wd_hi.w, cxv1, fmaf(wd_hi.z, cxv0, bs4.w + bd4.w));
        *reinterpret_cast<f32x4*>(cw + 4 * lane) = cv;
      }
      wave_sync_lds();
      f32x4 cinit[CT];   // C operand: destination term for channels ct*16 + 4g + r
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cinit[ct] = *reinterpret_cast<const f32x4*>(cw + ct * 16 + 4 * g);

      float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int base = 0; base < cdeg; base += 16) {
        const bool valid = base + j < cdeg;
        const float4 xc = xr;
        const float xB = xBn;
        {  // next tile of this destination, or the first tile of the next one: in flight while this tile computes
          const bool more = base + 16 < cdeg;
          request(more ? ce0 + base + 16 : n_e0, more ? cdeg - base - 16 : n_deg);
        }
        float e;
#define K1_CINIT_S(ct) cinit[ct]
        UAVGNN_TILE_SCORE(Wa, att, K1_CINIT_S, wlin, xB, k1_b_operand(xB, 0u), e)
        if (valid) {
          if (a_save_s != nullptr) {   // raw score now, weight once the maximum and the sum are final
            if (base < 16 * kStashTiles) srow[(base >> 4) * kWave + lane] = e;     // ... parked in the wavefront's row buffer
            else a_save_s[static_cast<size_t>(ce0 + base + j) * NH + g] = e;      // (more than 512 in-edges: through memory)
          }
          const float mn = fmaxf(m, e);
          const float sc = __builtin_amdgcn_exp2f(m - mn);   // exp2(-inf) = 0 on the first edge
          const float p = __builtin_amdgcn_exp2f(e - mn);
          den = fmaf(den, sc, p);
          s0 = fmaf(s0, sc, p * xc.x);
          s1 = fmaf(s1, sc, p * xc.y);
          s2 = fmaf(s2, sc, p * xc.z);
          s3 = fmaf(s3, sc, p * xc.w);
          m = mn;
        }
      }
      // ---- combine the 16 lanes of each head ----------------------------------------------------------------
      const float mx = row16_max(m);
      const float scl = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mx);
      const float inv = __builtin_amdgcn_rcpf(row16_sum(den * scl));
      f32x4 sv;
      sv[0] = row16_sum(s0 * scl) * inv;
      sv[1] = row16_sum(s1 * scl) * inv;
      sv[2] = row16_sum(s2 * scl) * inv;
      sv[3] = row16_sum(s3 * scl) * inv;
      if (a_save_s != nullptr) {
        // The raw scores of a destination's tiles wait in the per-wave row buffer of phase N (free in this phase; each lane
        // reads back what it wrote itself) instead of in a_save_s: written there and re-read for the normalisation they cost a
        // store, a DEPENDENT load - an L2 round trip at the end of every destination - and a second store per tile.
        for (int base = 0; base < cdeg; base += 16) {
          if (base + j < cdeg) {
            float* ap = a_save_s + static_cast<size_t>(ce0 + base + j) * NH + g;
            const float raw = base < 16 * kStashTiles ? srow[(base >> 4) * kWave + lane] : *ap;
            *ap = __builtin_amdgcn_exp2f(raw - mx) * inv;
          }
        }
      }
      // ---- epilogue: lane <-> 4 consecutive channels of head g; every lane of the row holds sv already -------
      float4 o;
      float* op = reinterpret_cast<float*>(&o);
      const float4 bs4 = reinterpret_cast<const float4*>(sBss)[lane], br4 = reinterpret_cast<const float4*>(sBrs)[lane];
      const float4 wr_lo = reinterpret_cast<const float4*>(sWrs)[2 * lane], wr_hi = reinterpret_cast<const float4*>(sWrs)[2 * lane + 1];
      const float bs[4] = {bs4.x, bs4.y, bs4.z, bs4.w};
      const float res[4] = {fmaf(wr_lo.y, cxv1, fmaf(wr_lo.x, cxv0, br4.x)), fmaf(wr_lo.w, cxv1, fmaf(wr_lo.z, cxv0, br4.y)),
                            fmaf(wr_hi.y, cxv1, fmaf(wr_hi.x, cxv0, br4.z)), fmaf(wr_hi.w, cxv1, fmaf(wr_hi.z, cxv0, br4.w))};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sWs + (4 * lane + r) * FS_S);
        float agg = bs[r];
        agg = fmaf(w[0], sv[0], agg);
        agg = fmaf(w[1], sv[1], agg);
        agg = fmaf(w[2], sv[2], agg);
        agg = fmaf(w[3], sv[3], agg);
        op[r] = fmaxf(agg + res[r], 0.f);
      }
      *reinterpret_cast<float4*>(out + static_cast<size_t>(v) * ld_out + 4 * lane) = o;
      if (rm_on) {
        float t = row16_max(fmaxf(fmaxf(op[0], op[1]), fmaxf(op[2], op[3])));
        t = fmaxf(t, __shfl_xor(t, 16));
        t = fmaxf(t, __shfl_xor(t, 32));
        if (lane == 0) rm_seen[v] = t;
      }
      wave_sync_lds();   // cw is rewritten by the next destination
    };

    // Destination meta data 64 hand-out positions at a time by VECTOR loads (lane <-> position) + v_readlane.  Scalar
    // loads would share the lgkm counter with the LDS traffic of process(): its first s_waitcnt lgkmcnt(0) would wait for
    // the look-ahead s_loads of the NEXT destination (an L2 round trip per destination).
    bool done = false;
    for (int kb = 0; !done && it0s + kb * stride < N; kb += kWave) {
      int m_v, m_e0, m_e1;
      float2 m_xv;
      if (kb == 0) {   // requested in the prologue
        const int* st = reinterpret_cast<const int*>(sC[wave]) + 2 * kWave;
        m_v = st[lane];
        m_e0 = st[kWave + lane];
        m_e1 = sS[wave][lane];
        m_xv = make_float2(__int_as_float(sS[wave][kWave + lane]), __int_as_float(sS[wave][2 * kWave + lane]));
      } else {
        const int my_it = it0s + (kb + lane) * stride;
        const bool mine = my_it < N;
        m_v = mine ? (seen_order ? seen_order[my_it] : my_it) : 0;
        m_e0 = mine ? seen_off[m_v] : 0;
        m_e1 = mine ? seen_off[m_v + 1] : 0;
        m_xv = mine ? *reinterpret_cast<const float2*>(x_dst + 2 * m_v) : make_float2(0.f, 0.f);
      }
      const int cnt = min(kWave, (N - it0s - kb * stride + stride - 1) / stride);
      {
        const int e0 = __builtin_amdgcn_readlane(m_e0, 0);
        request(e0, __builtin_amdgcn_readlane(m_e1, 0) - e0);
      }
      for (int ii = 0; ii < cnt; ++ii) {
        const int ce0 = __builtin_amdgcn_readlane(m_e0, ii);
        const int cdeg = __builtin_amdgcn_readlane(m_e1, ii) - ce0;
        if (cdeg == 0 && seen_order != nullptr) {   // sorted by decreasing degree: only isolated destinations are left
          done = true;
          break;
        }
        const int nx = min(ii + 1, kWave - 1);
        const int n_e0 = __builtin_amdgcn_readlane(m_e0, nx);
        const int n_deg = (ii + 1 < cnt) ? __builtin_amdgcn_readlane(m_e1, nx) - n_e0 : 0;
        if (cdeg == 0) {          // unordered hand-out: isolated destinations belong to phase N
          request(n_e0, n_deg);
          continue;
        }
        process(__builtin_amdgcn_readlane(m_v, ii), ce0, cdeg, rl(m_xv.x, ii), rl(m_xv.y, ii), n_e0, n_deg);
      }
    }
  }
  K1_STAMP(7)
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_gatv2_hetero_supported(int F_seen, int F_near, int F_dst, int nh, int D) {
  return (F_seen == FS_S && F_near == FS_N && F_dst == 2 && nh == NH && D == ::uavgnn::D) ? 1 : 0;
}

static int check_params(const float* const* seen_params, const float* const* near_params) {
  if (!seen_params || !near_params) return UAVGNN_EINVAL;
  for (int i = 0; i < 6; ++i)
    if (!seen_params[i] || !near_params[i]) return UAVGNN_EINVAL;
  for (int i = 0; i < 7; ++i)   // parameters are fetched with 16-byte loads
    if ((reinterpret_cast<uintptr_t>(seen_params[i]) & 15) || (reinterpret_cast<uintptr_t>(near_params[i]) & 15))
      return UAVGNN_EUNSUPPORTED;
  return 0;
}

static int k1_launch(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order, const float* x_ubs,
                     int E_near, const int32_t* near_off, const float* x_dst, int N, const float* const* seen_params,
                     const float* const* near_params, int nh, int D_, float slope, float* out, int ld_out,
                     float* attn_save_seen, float* attn_save_near, const void* image, int phases, uavgnn_stream_t stream,
                     float* rm_near = nullptr, float* rm_seen = nullptr) {
  if ((rm_near == nullptr) != (rm_seen == nullptr)) return UAVGNN_EINVAL;
  if (N < 0 || E_seen < 0 || E_near < 0 || (E_seen > 0 && !x_gt) || (E_near > 0 && !x_ubs) || !seen_off || !near_off ||
      !x_dst || !out || ld_out < 2 * nh * D_)
    return UAVGNN_EINVAL;
  if (!uavgnn_gatv2_hetero_supported(FS_S, FS_N, 2, nh, D_) || (ld_out & 3) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(x_gt) & 15) ||
      (reinterpret_cast<uintptr_t>(x_ubs) & 7) || (reinterpret_cast<uintptr_t>(x_dst) & 7) ||
      (reinterpret_cast<uintptr_t>(image) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (const int rc = check_params(seen_params, near_params)) return rc;
  if (N == 0) return 0;
  if (E_near == 0) x_ubs = x_dst;   // masked slots read row 0 of x_ubs: any valid address will do when there are no edges
  hipStream_t st = static_cast<hipStream_t>(stream);
#if !defined(K1_STANDALONE)
  if ((phases & 256) && rm_near != nullptr) return UAVGNN_EUNSUPPORTED;   // the fp32-MFMA build writes no row maxima
  if (phases & 256)
    return gatv2_hetero_launch_f32(x_gt, E_seen, seen_off, seen_order, x_ubs, near_off, x_dst, N, seen_params, near_params,
                                   slope, out, ld_out, attn_save_seen, attn_save_near, phases, st);
#endif
  RelParams ps{seen_params[0], seen_params[1], seen_params[2], seen_params[3], seen_params[4], seen_params[5], seen_params[6]};
  RelParams pn{near_params[0], near_params[1], near_params[2], near_params[3], near_params[4], near_params[5], near_params[6]};
  int grid = capped_grid(N, kWavesPerBlock, 256);   // persistent: one workgroup of eight wavefronts per CU
#if K1_ABLATE
  if (const char* gs = getenv("K1_GRID")) grid = atoi(gs);
#endif
  const k1_u32x4* img = static_cast<const k1_u32x4*>(image);
  const int ph = K1_ABLATE ? phases : (phases & 3);
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, st, x_gt, seen_off, seen_order, x_ubs, near_off, x_dst, N, E_seen, ps,
                       pn, slope, out, ld_out, attn_save_seen, attn_save_near, ph, img, rm_near, rm_seen);
  };
  const bool save = attn_save_near != nullptr || (attn_save_seen != nullptr && !(K1_ABLATE && (phases & 1024)));
  if (rm_near != nullptr) {
    if (save) {
      if (img != nullptr) launch(gatv2_hetero_fwd_kernel<true, true, true>); else launch(gatv2_hetero_fwd_kernel<true, false, true>);
    } else {
      if (img != nullptr) launch(gatv2_hetero_fwd_kernel<false, true, true>); else launch(gatv2_hetero_fwd_kernel<false, false, true>);
    }
  } else if (save) {
    if (img != nullptr) launch(gatv2_hetero_fwd_kernel<true, true>); else launch(gatv2_hetero_fwd_kernel<true, false>);
  } else {
    if (img != nullptr) launch(gatv2_hetero_fwd_kernel<false, true>); else launch(gatv2_hetero_fwd_kernel<false, false>);
  }
  return launch_status();
}

// phases: bit 0 = phase S, bit 1 = phase N (3 = the kernel; 1 / 2 are benchmark ablations, tools/kbench_hetero.py);
// bit 8 (UAVGNN_K1_FP32_MFMA) = the round 1-3 kernel with the score GEMM on fp32 MFMA (csrc/gatv2_hetero_f32.hip).
extern "C" int uavgnn_gatv2_hetero_fwd_phases(const float* x_gt, int E_seen, const int32_t* seen_off,
                                              const int32_t* seen_order, const float* x_ubs, int E_near,
                                              const int32_t* near_off, const float* x_dst, int N,
                                              const float* const* seen_params, const float* const* near_params, int nh,
                                              int D_, float slope, float* out, int ld_out, float* attn_save_seen,
                                              float* attn_save_near, int phases, uavgnn_stream_t stream) {
  return k1_launch(x_gt, E_seen, seen_off, seen_order, x_ubs, E_near, near_off, x_dst, N, seen_params, near_params, nh, D_,
                   slope, out, ld_out, attn_save_seen, attn_save_near, nullptr, phases, stream);
}

extern "C" size_t uavgnn_gatv2_hetero_image_bytes(void) { return sizeof(K1Image); }

// The parameter image of the two modules (K1Image), built once for any number of uavgnn_gatv2_hetero_fwd_image launches with the
// same parameter VALUES and slope.  The caller owns the buffer (16-byte aligned, uavgnn_gatv2_hetero_image_bytes()) and its
// validity: an image is stale the moment a parameter changes.
extern "C" int uavgnn_gatv2_hetero_prepare(const float* const* seen_params, const float* const* near_params, int nh, int D_,
                                           float slope, void* image, uavgnn_stream_t stream) {
  if (!image) return UAVGNN_EINVAL;
  if (!uavgnn_gatv2_hetero_supported(FS_S, FS_N, 2, nh, D_) || (reinterpret_cast<uintptr_t>(image) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (const int rc = check_params(seen_params, near_params)) return rc;
  RelParams ps{seen_params[0], seen_params[1], seen_params[2], seen_params[3], seen_params[4], seen_params[5], seen_params[6]};
  RelParams pn{near_params[0], near_params[1], near_params[2], near_params[3], near_params[4], near_params[5], near_params[6]};
  hipLaunchKernelGGL(gatv2_hetero_prepare_kernel, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), ps, pn, slope,
                     static_cast<k1_u32x4*>(image));
  return launch_status();
}

// uavgnn_gatv2_hetero_fwd with the parameters taken from a prepared image (the parameter arrays are still required: they select
// nothing on this path but keep the fp32-MFMA A/B route and the argument checks identical).  Bit-identical results.
extern "C" int uavgnn_gatv2_hetero_fwd_image(const float* x_gt, int E_seen, const int32_t* seen_off,
                                             const int32_t* seen_order, const float* x_ubs, int E_near,
                                             const int32_t* near_off, const float* x_dst, int N,
                                             const float* const* seen_params, const float* const* near_params, int nh,
                                             int D_, float slope, const void* image, float* out, int ld_out,
                                             float* attn_save_seen, float* attn_save_near, int phases,
                                             uavgnn_stream_t stream) {
  if (!image) return UAVGNN_EINVAL;
  return k1_launch(x_gt, E_seen, seen_off, seen_order, x_ubs, E_near, near_off, x_dst, N, seen_params, near_params, nh, D_,
                   slope, out, ld_out, attn_save_seen, attn_save_near, image, phases, stream);
}

// uavgnn_gatv2_hetero_fwd_image (image may be NULL: the in-kernel prologue) that ALSO writes rowmax_near / rowmax_seen [N]: the
// maximum over the `near` / `seen` half of every output row (rows are >= 0) - the two row bounds uavgnn_gemm_nt_h2 takes for the
// f_aggr product behind this launch (gnn_agents.py:106).  Not with the fp32-MFMA build (phases bit 8): UAVGNN_EUNSUPPORTED.
extern "C" int uavgnn_gatv2_hetero_fwd_rowmax(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                                              const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                                              const float* const* seen_params, const float* const* near_params, int nh, int D_,
                                              float slope, const void* image, float* out, int ld_out, float* attn_save_seen,
                                              float* attn_save_near, float* rowmax_near, float* rowmax_seen, int phases,
                                              uavgnn_stream_t stream) {
  if (!rowmax_near || !rowmax_seen) return UAVGNN_EINVAL;
  return k1_launch(x_gt, E_seen, seen_off, seen_order, x_ubs, E_near, near_off, x_dst, N, seen_params, near_params, nh, D_, slope, out,
                   ld_out, attn_save_seen, attn_save_near, image, phases, stream, rowmax_near, rowmax_seen);
}

extern "C" int uavgnn_gatv2_hetero_fwd(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                                       const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                                       const float* const* seen_params, const float* const* near_params, int nh, int D_,
                                       float slope, float* out, int ld_out, float* attn_save_seen,
                                       float* attn_save_near, uavgnn_stream_t stream) {
  return uavgnn_gatv2_hetero_fwd_phases(x_gt, E_seen, seen_off, seen_order, x_ubs, E_near, near_off, x_dst, N, seen_params,
                                        near_params, nh, D_, slope, out, ld_out, attn_save_seen, attn_save_near, 3, stream);
}
