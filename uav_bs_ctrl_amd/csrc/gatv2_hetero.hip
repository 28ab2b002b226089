// K1 forward of the WHOLE observation encoder in ONE launch: both GATv2 relations of
// /root/reference/algos/madrqn/agents/gnn_agents.py:93-96,:103-104 (`gt -seen-> agent`, F = 4 and `ubs -near-> agent`,
// F = 2; 4 heads x 64 channels; math SURVEY Appendix A.1/A.3) write the two halves of one [N, 2H] row (the th.cat of
// gnn_agents.py:106 never exists), with one constant-load prologue and one pass over the destination meta data.
//
// Every persistent wavefront runs two phases:
//
//  S  the `seen` relation of the destinations that HAVE in-edges, one destination at a time on 16-edge row tiles - the
//     formulation of gatv2_mfma.hip (Z^T = W_s X^T + c[v] on the matrix cores, |z| half of the leaky ReLU as one
//     |.|-modifier FMA per channel, permlane head reduction, log2-domain online softmax, input-space aggregation) with
//     three changes: lane <-> FOUR CONSECUTIVE channels in the per-destination prologue / epilogue (one 16-byte LDS
//     write, five 16-byte LDS reads and one 16-byte row store instead of 4 + 32 + 4 dword operations), the first row tile
//     of the NEXT destination is requested while the last tile of the current one computes (no exposed HBM latency at
//     a destination boundary), and destinations without in-edges are left to phase N.  With a hand-out order (sorted
//     by decreasing degree) the phase ends at the first isolated destination - 94 % of the agents of a random-policy
//     rollout never enter it.
//
//  N  the `near` relation of ALL destinations, TWO destinations per row tile, plus the residual-only `seen` half of the
//     isolated ones.  A `near` edge has two source features, so the K = 4 contraction of the MFMA holds
//     [x_u ; x_v]: A = [W_s | W_d] rows, B = (x_u0, x_u1, x_v0, x_v1) per column, C = b_s + b_d - the destination term
//     needs no per-destination C operand, so the 16 columns of a tile may belong to different destinations: columns
//     0-7 carry the in-edges of destination 2p, columns 8-15 those of 2p+1 (degrees above 8 take further passes with
//     the online softmax).  The segment softmax is an all-reduce over 8 lanes (two quad permutes + row_half_mirror),
//     nothing crosses a 16-lane row, no LDS in the tile loop.  n - 1 = 7 neighbours -> 7/8 of the columns do work,
//     where one destination per tile wastes 9/16 and the lane <-> channel kernel (gatv2_small.hip) spends ~350 VALU
//     instructions per destination.
//
// Arithmetic: fp32 in / out / accumulate.  The score GEMM runs on the bf16 matrix cores as six exact bf16 products per fp32
// product (K1_BF16Z, below): results equal the per-relation fp32-MFMA kernels' to fp32 rounding (1e-9 relative on the output
// checksum), not bit for bit.
// Instantiated for D = 64 (H = 256: every BASELINE configuration); other shapes use the per-relation kernels.
#include "common.h"

namespace uavgnn {
namespace {

#ifndef K1_BF16Z
#define K1_BF16Z 1    // 1: the score GEMM z = W x + c on v_mfma_f32_16x16x32_bf16 with exact three-way operand splits; 0: fp32 MFMA
#endif
#ifndef K1_ABLATE
#define K1_ABLATE 0   // 1 (tools/ubench/k1_env_bench.hip only): `phases` bit 5 skips the score tile of phase N, bit 6 its row stores
#endif
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int NH = 4;
constexpr int D = 64;
constexpr int H = NH * D;          // 256
constexpr int CT = H / 16;         // 16 channel tiles
constexpr int TPH = D / 16;        // channel tiles per head
constexpr int FS_S = 4, FS_N = 2;
constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
constexpr int kRowRor = 0x120;        // row_ror:n
constexpr int kQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kHalfMirror = 0x141;    // row_half_mirror: lane i <-> 7 - i inside every 8 lanes

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
// all-reduce over the 8 lanes of a half row (the in-edge slots of ONE destination in phase N)
__device__ __forceinline__ float half8_sum(float v) {
  v += dpp_mov<kQuadXor1>(v);
  v += dpp_mov<kQuadXor2>(v);
  v += dpp_mov<kHalfMirror>(v);
  return v;
}
__device__ __forceinline__ float half8_max(float v) {
  v = fmaxf(v, dpp_mov<kQuadXor1>(v));
  v = fmaxf(v, dpp_mov<kQuadXor2>(v));
  v = fmaxf(v, dpp_mov<kHalfMirror>(v));
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

__device__ __forceinline__ float rl(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

struct RelParams {   // one GATv2Conv: fc_src, fc_dst, attn, res_fc (DGL layout, gnn_agents.py:93-96)
  const float* W_s; const float* b_s; const float* W_d; const float* b_d; const float* attn; const float* W_r;
  const float* b_r;
};

// The MFMA + |z| FMA block shared by both phases: 16 channel tiles of one row tile -> log2-domain score of
// (column j, head g) in every lane, up to a per-(destination, head) constant that cancels in the softmax.
// The instruction order is PINNED (sched_barrier): MFMA ct+1 is issued, then the four |z| FMAs of tile ct run in its
// shadow.  Left to itself hipcc hoists / sinks the FMAs around the MFMAs and the tile takes ~10 % longer
// (tools/ubench/tile_sched.hip: 510 vs 455-463 ns per tile per SIMD at two waves per SIMD).
#if K1_BF16Z
// ---- the score GEMM on the bf16 matrix cores -------------------------------------------------------------------------
// fp32 MFMA on gfx950 issues at the fp32 VECTOR rate and does not overlap the VALU work of the same SIMD (measured additive:
// profiles/r03_k1_env_ablation.txt, tools/ubench/tile_sched.hip), so the 16 fp32 MFMAs of a row tile cost as much as its
// ~105 VALU instructions.  The K = 4 contraction is therefore laid out over the K = 32 of ONE v_mfma_f32_16x16x32_bf16 per
// channel tile: K group g (8 slots, = lane group g on both operands) holds the six bf16 x bf16 products of feature g -
//     A = (w1 w1 | w2 w2 | w1 w3 | 0 0)      B = (x1 x2 | x1 x2 | x3 x1 | 0 0)      w = w1 + w2 + w3, x = x1 + x2 + x3 exactly
// (bf16x3.h: every product exact in the fp32 accumulator, the three dropped ones <= 2^-23 |w x|: one fp32 rounding).  A lane
// owns ONE feature of its edge (the value it already loads as the fp32 B operand), so the B operand is one three-way split per
// tile (9 VALU); the A operands are split once per wavefront (64 VGPRs instead of 16).  The destination term stays the fp32 C
// operand.  tools/ubench/tile_sched.hip: 354 ns per tile per SIMD against 454 for the fp32 MFMA tile (2 waves per SIMD).
typedef __bf16 k1_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 k1_bf16x8 __attribute__((ext_vector_type(8)));
typedef float k1_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned k1_u32x4 __attribute__((ext_vector_type(4)));
struct K1Split { unsigned h1, h2, h3; };   // (t1 | t1 << 16) of the three terms
__device__ __forceinline__ K1Split k1_split(float x) {
  K1Split s;
  k1_bf16x2 p = __builtin_convertvector(k1_f32x2{x, x}, k1_bf16x2);
  s.h1 = __builtin_bit_cast(unsigned, p);
  const float r1 = x - __uint_as_float(s.h1 & 0xffff0000u);
  p = __builtin_convertvector(k1_f32x2{r1, r1}, k1_bf16x2);
  s.h2 = __builtin_bit_cast(unsigned, p);
  const float r2 = r1 - __uint_as_float(s.h2 & 0xffff0000u);
  p = __builtin_convertvector(k1_f32x2{r2, r2}, k1_bf16x2);
  s.h3 = __builtin_bit_cast(unsigned, p);
  return s;
}
__device__ __forceinline__ k1_u32x4 k1_a_operand(float w) {   // (w1 w1 | w2 w2 | w1 w3 | 0 0)
  const K1Split s = k1_split(w);
  return k1_u32x4{s.h1, s.h2, (s.h1 & 0xffffu) | (s.h3 & 0xffff0000u), 0u};
}
__device__ __forceinline__ k1_bf16x8 k1_b_operand(float x) {   // (x1 x2 | x1 x2 | x3 x1 | 0 0)
  const K1Split s = k1_split(x);
  const unsigned x12 = (s.h1 & 0xffffu) | (s.h2 & 0xffff0000u);
  return __builtin_bit_cast(k1_bf16x8, k1_u32x4{x12, x12, (s.h3 & 0xffffu) | (s.h1 & 0xffff0000u), 0u});
}
#define K1_WA_T k1_u32x4
#define K1_WA_INIT(w) k1_a_operand(w)
#define K1_MFMA(WA_ct, XBOP, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k1_bf16x8, WA_ct), XBOP, C, 0, 0, 0)
#define K1_XB_OP(xb) k1_b_operand(xb)
// phase N: the channel bias b_s + b_d rides in the free K slots (A: b1 b2 | b3 0 in the last word of lane groups 0 / 1, B: 1 1
// | 1 0 there), so the C operand is the inline constant 0 and no register holds it
__device__ __forceinline__ k1_u32x4 k1_a_operand_bias(float w, float bias, int g) {
  k1_u32x4 a = k1_a_operand(w);
  const K1Split b = k1_split(bias);
  a[3] = g == 0 ? ((b.h1 & 0xffffu) | (b.h2 & 0xffff0000u)) : g == 1 ? (b.h3 & 0xffffu) : 0u;
  return a;
}
__device__ __forceinline__ k1_bf16x8 k1_b_operand_one(float x, unsigned one_word) {
  const K1Split s = k1_split(x);
  const unsigned x12 = (s.h1 & 0xffffu) | (s.h2 & 0xffff0000u);
  return __builtin_bit_cast(k1_bf16x8, k1_u32x4{x12, x12, (s.h3 & 0xffffu) | (s.h1 & 0xffff0000u), one_word});
}
#else
#define K1_WA_T float
#define K1_WA_INIT(w) (w)
#define K1_MFMA(WA_ct, XBOP, C) __builtin_amdgcn_mfma_f32_16x16x4f32(WA_ct, XBOP, C, 0, 0, 0)
#define K1_XB_OP(xb) (xb)
#endif
#define UAVGNN_TILE_SCORE(WA, ATT, CINIT, WLIN, XB, E_OUT) UAVGNN_TILE_SCORE_(WA, ATT, CINIT, WLIN, XB, K1_XB_OP(XB), E_OUT)
#define K1_INDEX(A) A
#define UAVGNN_TILE_SCORE_(WA, ATT, CINIT, WLIN, XB, XBOP, E_OUT)                                       \
  {                                                                                                     \
    float pe[NH][2];                                                                                    \
    _Pragma("unroll") for (int k = 0; k < NH; ++k) {                                                    \
      pe[k][0] = WLIN[k] * (XB);                                                                        \
      pe[k][1] = 0.f;                                                                                   \
    }                                                                                                   \
    const auto xb_op = XBOP;                                                                            \
    f32x4 z_cur = K1_MFMA(WA[0], xb_op, CINIT(0));                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                 \
      f32x4 z_nxt = z_cur;                                                                              \
      if (ct + 1 < CT) {                                                                                \
        z_nxt = K1_MFMA(WA[ct + 1], xb_op, CINIT(ct + 1));                                              \
        __builtin_amdgcn_sched_barrier(0);                                                              \
      }                                                                                                 \
      const int k = ct / TPH;                                                                           \
      pe[k][0] = fmaf(ATT[ct][0], fabsf(z_cur[0]), pe[k][0]);                                           \
      pe[k][1] = fmaf(ATT[ct][1], fabsf(z_cur[1]), pe[k][1]);                                           \
      pe[k][0] = fmaf(ATT[ct][2], fabsf(z_cur[2]), pe[k][0]);                                           \
      pe[k][1] = fmaf(ATT[ct][3], fabsf(z_cur[3]), pe[k][1]);                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      z_cur = z_nxt;                                                                                    \
    }                                                                                                   \
    E_OUT = reduce_heads(pe[0][0] + pe[0][1], pe[1][0] + pe[1][1], pe[2][0] + pe[2][1], pe[3][0] + pe[3][1]); \
  }

__global__ __launch_bounds__(kThreads, 2) void gatv2_hetero_fwd_kernel(
    const float* __restrict__ x_gt, const int32_t* __restrict__ seen_off, const int32_t* __restrict__ seen_order,
    const float* __restrict__ x_ubs, const int32_t* __restrict__ near_off, const float* __restrict__ x_dst, int N,
    int E_seen, RelParams ps, RelParams pn, float slope, float* __restrict__ out, int ld_out, float* __restrict__ a_save_s,
    float* __restrict__ a_save_n, int phases) {
  __shared__ __attribute__((aligned(16))) float sWs[H * FS_S];     // fc_src.weight of `seen`, row-major [H, 4]
  __shared__ __attribute__((aligned(16))) float sWn[H * FS_N];     // fc_src.weight of `near`, row-major [H, 2]
  __shared__ __attribute__((aligned(16))) float sAs[H], sAn[H];     // attention vectors
  __shared__ __attribute__((aligned(16))) float sWdn[H * 2];        // fc_dst.weight of `near` (A operand rows of phase N)
  __shared__ __attribute__((aligned(16))) float sBCn[H];            // b_s + b_d of `near` (the constant C operand)
  // the remaining small operands, staged once per workgroup instead of fetched by every wavefront:
  __shared__ __attribute__((aligned(16))) float sWds[H * 2], sWrs[H * 2], sWrn[H * 2];   // seen fc_dst / res_fc, near res_fc
  __shared__ __attribute__((aligned(16))) float sBss[H], sBds[H], sBrs[H], sBsn[H], sBrn[H];   // biases (b_r: 0 when absent)
  __shared__ float sWa[2][NH * 4];                                  // wa[k][f] = sum_d attn[k,d] W_s[k,d,f] per relation
  __shared__ __attribute__((aligned(16))) float sC[kWavesPerBlock][H];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15;          // MFMA column (edge slot) / A row
  const int g = lane >> 4;          // lane group: K index of the A / B operands, head after the reduction

  for (int i = tid; i < H * FS_S / 4; i += kThreads)
    reinterpret_cast<float4*>(sWs)[i] = reinterpret_cast<const float4*>(ps.W_s)[i];
  for (int i = tid; i < H * FS_N / 4; i += kThreads) {
    reinterpret_cast<float4*>(sWn)[i] = reinterpret_cast<const float4*>(pn.W_s)[i];
    reinterpret_cast<float4*>(sWdn)[i] = reinterpret_cast<const float4*>(pn.W_d)[i];
    reinterpret_cast<float4*>(sWds)[i] = reinterpret_cast<const float4*>(ps.W_d)[i];
    reinterpret_cast<float4*>(sWrs)[i] = reinterpret_cast<const float4*>(ps.W_r)[i];
    reinterpret_cast<float4*>(sWrn)[i] = reinterpret_cast<const float4*>(pn.W_r)[i];
  }
  for (int i = tid; i < H; i += kThreads) {
    sAs[i] = ps.attn[i];
    sAn[i] = pn.attn[i];
    sBCn[i] = pn.b_s[i] + pn.b_d[i];
    sBss[i] = ps.b_s[i];
    sBds[i] = ps.b_d[i];
    sBsn[i] = pn.b_s[i];
    sBrs[i] = ps.b_r != nullptr ? ps.b_r[i] : 0.f;
    sBrn[i] = pn.b_r != nullptr ? pn.b_r[i] : 0.f;
  }
  __syncthreads();
  {  // wa[rel][k][f]: 2 x 16 outputs, 16 partial sums each; 256 threads = 16 rows of 16 lanes, two rounds
    const int kf = tid >> 4, part = tid & 15;
    const int k = kf >> 2, f = kf & 3;
    float a0 = 0.f, a1 = 0.f;
    for (int d = part; d < D; d += 16) {
      a0 = fmaf(sAs[k * D + d], sWs[(k * D + d) * FS_S + f], a0);
      if (f < FS_N) a1 = fmaf(sAn[k * D + d], sWn[(k * D + d) * FS_N + f], a1);
    }
    a0 = row16_sum(a0);
    a1 = row16_sum(a1);
    if (part == 0) {
      sWa[0][kf] = a0;
      sWa[1][kf] = a1;
    }
  }
  __syncthreads();

  const float c_abs = kLog2e * 0.5f * (1.f - slope), c_lin = kLog2e * 0.5f * (1.f + slope);
  float* __restrict__ cw = sC[wave];
  const int stride = gridDim.x * kWavesPerBlock;
  const int it0 = blockIdx.x * kWavesPerBlock + wave;

  // =========================== phase S: `seen` on the destinations that have in-edges ===============================
  if (it0 < N && (phases & 1) && E_seen > 0) {
    K1_WA_T Wa[CT];
    float att[CT][4], wlin[NH];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      Wa[ct] = K1_WA_INIT(sWs[(ct * 16 + j) * FS_S + g]);
#pragma unroll
      for (int r = 0; r < 4; ++r) att[ct][r] = c_abs * sAs[ct * 16 + 4 * g + r];
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) wlin[k] = c_lin * sWa[0][k * 4 + g];
    // The per-destination constants of the prologue / epilogue (lane <-> channels 4*lane .. 4*lane+3: fc_dst, res_fc rows,
    // biases) are READ FROM LDS where they are used, once per destination: the bf16 A operands take 64 VGPRs (K1_BF16Z) and
    // 28 more registers held across the tile loop would spill.
    // inputs of the row tile that is processed next (possibly the first tile of the next destination)
    float4 xr;
    float xBn;
    // ALWAYS exactly two loads (clamped address, never predicated): a predicated load makes the number of outstanding
    // loads unknown to the compiler, which then drains the queue (s_waitcnt vmcnt(0)) right behind the request - the
    // prefetch would not overlap anything.  Columns j >= count read edge row 0; they are masked out of the softmax.
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
        UAVGNN_TILE_SCORE(Wa, att, K1_CINIT_S, wlin, xB, e)
        if (valid) {
          if (a_save_s != nullptr) a_save_s[static_cast<size_t>(ce0 + base + j) * NH + g] = e;
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
        for (int base = 0; base < cdeg; base += 16) {
          if (base + j < cdeg) {
            float* ap = a_save_s + static_cast<size_t>(ce0 + base + j) * NH + g;
            *ap = __builtin_amdgcn_exp2f(*ap - mx) * inv;
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
      wave_sync_lds();   // cw is rewritten by the next destination
    };

    // Destination meta data 64 hand-out positions at a time by VECTOR loads (lane <-> position) + v_readlane.  Scalar
    // loads would share the lgkm counter with the LDS traffic of process(): its first s_waitcnt lgkmcnt(0) would wait for
    // the look-ahead s_loads of the NEXT destination (an L2 round trip per destination - what the per-relation kernel
    // pays: ~1.5k cycles per destination).
    bool done = false;
    for (int kb = 0; !done && it0 + kb * stride < N; kb += kWave) {
      const int my_it = it0 + (kb + lane) * stride;
      const bool mine = my_it < N;
      const int m_v = mine ? (seen_order ? seen_order[my_it] : my_it) : 0;
      const int m_e0 = mine ? seen_off[m_v] : 0;
      const int m_e1 = mine ? seen_off[m_v + 1] : 0;
      const float2 m_xv = mine ? *reinterpret_cast<const float2*>(x_dst + 2 * m_v) : make_float2(0.f, 0.f);
      const int cnt = min(kWave, (N - it0 - kb * stride + stride - 1) / stride);
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

  // ================= phase N: `near`, two destinations per row tile, + residual-only `seen` rows ====================
  if (phases & 2) {
    K1_WA_T Wa[CT];
    float att[CT][4], wlin[NH];
#if K1_BF16Z
    const unsigned one_word = g == 0 ? 0x3F803F80u : g == 1 ? 0x00003F80u : 0u;   // bf16 1.0 against the bias slots of the A operand
    const f32x4 czero = {0.f, 0.f, 0.f, 0.f};
#define K1_CINIT_N(ct) czero
#define K1_XBOP_N(xb) k1_b_operand_one(xb, one_word)
#else
    f32x4 cconst[CT];
#define K1_CINIT_N(ct) cconst[ct]
#define K1_XBOP_N(xb) (xb)
#endif
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int row = ct * 16 + j;
      const float wv = (g < 2) ? sWn[row * FS_N + g] : sWdn[row * 2 + (g - 2)];
#if K1_BF16Z
      Wa[ct] = k1_a_operand_bias(wv, sBCn[row], g);
#else
      Wa[ct] = wv;
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = ct * 16 + 4 * g + r;
        att[ct][r] = c_abs * sAn[ch];
#if !K1_BF16Z
        cconst[ct][r] = sBCn[ch];
#endif
      }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) wlin[k] = (g < 2) ? c_lin * sWa[1][k * 4 + g] : 0.f;
    // epilogue constants, lane <-> channels 4*lane .. 4*lane+3
#define K1_EPI_CONSTS                                                                                                              \
      float ws[4][2], bsn[4], wrn[4][2], brn[4], wrs[4][2], brs[4];                                                                \
      {                                                                                                                            \
        const float4 bs4 = reinterpret_cast<const float4*>(sBsn)[lane];                                                            \
        const float4 ws_lo = reinterpret_cast<const float4*>(sWn)[2 * lane], ws_hi = reinterpret_cast<const float4*>(sWn)[2 * lane + 1];   \
        const float4 wn_lo = reinterpret_cast<const float4*>(sWrn)[2 * lane], wn_hi = reinterpret_cast<const float4*>(sWrn)[2 * lane + 1]; \
        const float4 wq_lo = reinterpret_cast<const float4*>(sWrs)[2 * lane], wq_hi = reinterpret_cast<const float4*>(sWrs)[2 * lane + 1]; \
        const float4 bn4 = reinterpret_cast<const float4*>(sBrn)[lane], bq4 = reinterpret_cast<const float4*>(sBrs)[lane];          \
        bsn[0] = bs4.x; bsn[1] = bs4.y; bsn[2] = bs4.z; bsn[3] = bs4.w;                                                            \
        ws[0][0] = ws_lo.x; ws[0][1] = ws_lo.y; ws[1][0] = ws_lo.z; ws[1][1] = ws_lo.w;                                            \
        ws[2][0] = ws_hi.x; ws[2][1] = ws_hi.y; ws[3][0] = ws_hi.z; ws[3][1] = ws_hi.w;                                            \
        wrn[0][0] = wn_lo.x; wrn[0][1] = wn_lo.y; wrn[1][0] = wn_lo.z; wrn[1][1] = wn_lo.w;                                        \
        wrn[2][0] = wn_hi.x; wrn[2][1] = wn_hi.y; wrn[3][0] = wn_hi.z; wrn[3][1] = wn_hi.w;                                        \
        wrs[0][0] = wq_lo.x; wrs[0][1] = wq_lo.y; wrs[1][0] = wq_lo.z; wrs[1][1] = wq_lo.w;                                        \
        wrs[2][0] = wq_hi.x; wrs[2][1] = wq_hi.y; wrs[3][0] = wq_hi.z; wrs[3][1] = wq_hi.w;                                        \
        brn[0] = bn4.x; brn[1] = bn4.y; brn[2] = bn4.z; brn[3] = bn4.w;                                                            \
        brs[0] = bq4.x; brs[1] = bq4.y; brs[2] = bq4.z; brs[3] = bq4.w;                                                            \
      }
    K1_EPI_CONSTS
    const int half = j >> 3, slot = j & 7;
    const int P = (N + 1) >> 1;     // destination pairs (2p, 2p+1)

    // pair meta data 64 at a time by vector loads (lane <-> pair), handed over by v_readlane: no scalar-load chain
    for (int kb = 0; it0 + kb * stride < P; kb += kWave) {
      const int my_p = it0 + (kb + lane) * stride;
      const bool mine = my_p < P;
      const int vA = 2 * my_p;
      const bool hasB = mine && (vA + 1 < N);
      const int m_n0 = mine ? near_off[vA] : 0;
      const int m_n1 = mine ? near_off[vA + 1] : 0;
      const int m_n2 = hasB ? near_off[vA + 2] : m_n1;
      const int m_s0 = mine ? seen_off[vA] : 0;
      const int m_s1 = mine ? seen_off[vA + 1] : 0;
      const int m_s2 = hasB ? seen_off[vA + 2] : m_s1;
      const float2 m_xa = mine ? *reinterpret_cast<const float2*>(x_dst + 2 * vA) : make_float2(0.f, 0.f);
      const float2 m_xb = hasB ? *reinterpret_cast<const float2*>(x_dst + 2 * vA + 2) : make_float2(0.f, 0.f);
      // bit 0 / 1: destination A / B has no `seen` in-edge (its residual-only row is written here); bit 2: B exists
      const int m_flags = (m_s1 == m_s0 ? 1 : 0) | ((hasB && m_s2 == m_s1) ? 2 : 0) | (hasB ? 4 : 0);
      const int cnt = min(kWave, (P - it0 - kb * stride + stride - 1) / stride);
      // first-pass inputs of the NEXT pair are requested before the current pair computes
      int q_n0 = __builtin_amdgcn_readlane(m_n0, 0), q_n1 = __builtin_amdgcn_readlane(m_n1, 0);
      int q_n2 = __builtin_amdgcn_readlane(m_n2, 0);
      float2 xq;   // clamped, never predicated loads (see phase S); masked slots read edge row 0
      {
        const int e0 = half ? q_n1 : q_n0, dg = half ? q_n2 - q_n1 : q_n1 - q_n0;
        xq = *reinterpret_cast<const float2*>(x_ubs + static_cast<size_t>(slot < dg ? e0 + slot : 0) * FS_N);
      }
      for (int ii = 0; ii < cnt; ++ii) {
        const int p = it0 + (kb + ii) * stride;
        const int n0 = q_n0, n1 = q_n1, n2 = q_n2;
        float2 xu = xq;
        if (ii + 1 < cnt) {
          q_n0 = __builtin_amdgcn_readlane(m_n0, ii + 1);
          q_n1 = __builtin_amdgcn_readlane(m_n1, ii + 1);
          q_n2 = __builtin_amdgcn_readlane(m_n2, ii + 1);
          const int e0 = half ? q_n1 : q_n0, dg = half ? q_n2 - q_n1 : q_n1 - q_n0;
          xq = *reinterpret_cast<const float2*>(x_ubs + static_cast<size_t>(slot < dg ? e0 + slot : 0) * FS_N);
        }
        const int flags = __builtin_amdgcn_readlane(m_flags, ii);
        const float xa0 = rl(m_xa.x, ii), xa1 = rl(m_xa.y, ii), xb0 = rl(m_xb.x, ii), xb1 = rl(m_xb.y, ii);
        const int degA = n1 - n0, degB = n2 - n1;
        const int my_e0 = half ? n1 : n0, my_deg = half ? degB : degA;
        const float xv_g = (g == 2) ? (half ? xb0 : xa0) : (half ? xb1 : xa1);   // only read by lane groups 2, 3

        float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f;
        const int dmax = max(degA, degB);
        for (int base = 0; base < dmax; base += 8) {
          const bool valid = base + slot < my_deg;
          const float2 xc = xu;
          if (base + 8 < dmax)       // degrees above 8: further passes
            xu = *reinterpret_cast<const float2*>(
                x_ubs + static_cast<size_t>(base + 8 + slot < my_deg ? my_e0 + base + 8 + slot : 0) * FS_N);
          const float xB = (g == 0) ? xc.x : (g == 1) ? xc.y : xv_g;
          float e;
#if K1_ABLATE
          if (phases & 32) e = xB; else
#endif
          UAVGNN_TILE_SCORE_(Wa, att, K1_CINIT_N, wlin, xB, K1_XBOP_N(xB), e)
          if (valid) {
            if (a_save_n != nullptr) a_save_n[static_cast<size_t>(my_e0 + base + slot) * NH + g] = e;
            const float mn = fmaxf(m, e);
            const float sc = __builtin_amdgcn_exp2f(m - mn);
            const float pw = __builtin_amdgcn_exp2f(e - mn);
            den = fmaf(den, sc, pw);
            s0 = fmaf(s0, sc, pw * xc.x);
            s1 = fmaf(s1, sc, pw * xc.y);
            m = mn;
          }
        }
        // ---- segment softmax: all-reduce over the 8 slots of each destination ---------------------------------
        const float mx = half8_max(m);
        const float scl = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mx);
        const float dsum = half8_sum(den * scl);
        const float inv = dsum > 0.f ? __builtin_amdgcn_rcpf(dsum) : 0.f;      // isolated destination: aggregate = 0
        const float t0 = half8_sum(s0 * scl) * inv, t1 = half8_sum(s1 * scl) * inv;
        if (a_save_n != nullptr) {
          for (int base = 0; base < my_deg; base += 8) {
            if (base + slot < my_deg) {
              float* ap = a_save_n + static_cast<size_t>(my_e0 + base + slot) * NH + g;
              *ap = __builtin_amdgcn_exp2f(*ap - mx) * inv;
            }
          }
        }
        // head g's aggregated inputs of BOTH destinations in every lane of the row
        const float o0 = dpp_mov<kRowRor + 8>(t0), o1 = dpp_mov<kRowRor + 8>(t1);
        const float sA0 = half ? o0 : t0, sA1 = half ? o1 : t1, sB0 = half ? t0 : o0, sB1 = half ? t1 : o1;
        // ---- epilogue: lane <-> 4 consecutive channels of head g ----------------------------------------------
        float* rowA = out + static_cast<size_t>(2 * p) * ld_out;
        {
          float4 o;
          float* op = reinterpret_cast<float*>(&o);
          const float has = degA > 0 ? 1.f : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float agg = has * fmaf(ws[r][1], sA1, fmaf(ws[r][0], sA0, bsn[r]));
            op[r] = fmaxf(agg + fmaf(wrn[r][1], xa1, fmaf(wrn[r][0], xa0, brn[r])), 0.f);
          }
#if K1_ABLATE
          if (!(phases & 64) || o.x == 123.456f)
#endif
          *reinterpret_cast<float4*>(rowA + H + 4 * lane) = o;
          if (flags & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) op[r] = fmaxf(fmaf(wrs[r][1], xa1, fmaf(wrs[r][0], xa0, brs[r])), 0.f);
#if K1_ABLATE
            if (!(phases & 64) || o.x == 123.456f)
#endif
            *reinterpret_cast<float4*>(rowA + 4 * lane) = o;
          }
        }
        if (flags & 4) {
          float* rowB = rowA + ld_out;
          float4 o;
          float* op = reinterpret_cast<float*>(&o);
          const float has = degB > 0 ? 1.f : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float agg = has * fmaf(ws[r][1], sB1, fmaf(ws[r][0], sB0, bsn[r]));
            op[r] = fmaxf(agg + fmaf(wrn[r][1], xb1, fmaf(wrn[r][0], xb0, brn[r])), 0.f);
          }
#if K1_ABLATE
          if (!(phases & 64) || o.x == 123.456f)
#endif
          *reinterpret_cast<float4*>(rowB + H + 4 * lane) = o;
          if (flags & 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) op[r] = fmaxf(fmaf(wrs[r][1], xb1, fmaf(wrs[r][0], xb0, brs[r])), 0.f);
#if K1_ABLATE
            if (!(phases & 64) || o.x == 123.456f)
#endif
            *reinterpret_cast<float4*>(rowB + 4 * lane) = o;
          }
        }
      }
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

// The file is compiled twice: as itself (score GEMM on the bf16 matrix cores) and through gatv2_hetero_f32.hip
// (-> K1_BF16Z = 0, K1_F32_TU: the fp32-MFMA score GEMM of rounds 1-2, reachable as phases bit 8 = the A/B and strict-fp32 leg).
namespace uavgnn {
#if defined(K1_F32_TU)
int gatv2_hetero_launch_f32(
#else
int gatv2_hetero_launch_f32(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order, const float* x_ubs,
                            const int32_t* near_off, const float* x_dst, int N, const float* const* seen_params,
                            const float* const* near_params, float slope, float* out, int ld_out, float* attn_save_seen,
                            float* attn_save_near, int phases, hipStream_t st);
static int gatv2_hetero_launch(
#endif
    const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order, const float* x_ubs,
    const int32_t* near_off, const float* x_dst, int N, const float* const* seen_params, const float* const* near_params,
    float slope, float* out, int ld_out, float* attn_save_seen, float* attn_save_near, int phases, hipStream_t st) {
  RelParams ps{seen_params[0], seen_params[1], seen_params[2], seen_params[3], seen_params[4], seen_params[5], seen_params[6]};
  RelParams pn{near_params[0], near_params[1], near_params[2], near_params[3], near_params[4], near_params[5], near_params[6]};
  int grid = capped_grid(N, kWavesPerBlock, 512);   // persistent: 2 workgroups per CU
#if K1_ABLATE
  if (const char* gs = getenv("K1_GRID")) grid = atoi(gs);
#endif
  hipLaunchKernelGGL(gatv2_hetero_fwd_kernel, dim3(grid), dim3(kThreads), 0, st, x_gt, seen_off, seen_order, x_ubs, near_off,
                     x_dst, N, E_seen, ps, pn, slope, out, ld_out, attn_save_seen, attn_save_near,
                     K1_ABLATE ? phases : (phases & 3));
  return launch_status();
}
}  // namespace uavgnn

#if !defined(K1_F32_TU)
extern "C" int uavgnn_gatv2_hetero_supported(int F_seen, int F_near, int F_dst, int nh, int D) {
  return (F_seen == FS_S && F_near == FS_N && F_dst == 2 && nh == NH && D == ::uavgnn::D) ? 1 : 0;
}

// phases: bit 0 = phase S, bit 1 = phase N (3 = the kernel; 1 / 2 are benchmark ablations, tools/kbench_hetero.py);
// bit 8 (UAVGNN_K1_FP32_MFMA) = the score GEMM on fp32 MFMA instead of the bf16 matrix cores.
extern "C" int uavgnn_gatv2_hetero_fwd_phases(const float* x_gt, int E_seen, const int32_t* seen_off,
                                              const int32_t* seen_order, const float* x_ubs, int E_near,
                                              const int32_t* near_off, const float* x_dst, int N,
                                              const float* const* seen_params, const float* const* near_params, int nh,
                                              int D_, float slope, float* out, int ld_out, float* attn_save_seen,
                                              float* attn_save_near, int phases, uavgnn_stream_t stream) {
  if (N < 0 || E_seen < 0 || E_near < 0 || (E_seen > 0 && !x_gt) || (E_near > 0 && !x_ubs) || !seen_off || !near_off ||
      !x_dst || !seen_params || !near_params || !out || ld_out < 2 * nh * D_)
    return UAVGNN_EINVAL;
  if (!uavgnn_gatv2_hetero_supported(FS_S, FS_N, 2, nh, D_) || (ld_out & 3) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(x_gt) & 15) ||
      (reinterpret_cast<uintptr_t>(x_ubs) & 7) || (reinterpret_cast<uintptr_t>(x_dst) & 7))
    return UAVGNN_EUNSUPPORTED;
  for (int i = 0; i < 6; ++i)
    if (!seen_params[i] || !near_params[i]) return UAVGNN_EINVAL;
  for (int i = 0; i < 7; ++i)   // parameters are fetched with 16-byte loads
    if ((reinterpret_cast<uintptr_t>(seen_params[i]) & 15) || (reinterpret_cast<uintptr_t>(near_params[i]) & 15))
      return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  if (E_near == 0) x_ubs = x_dst;   // masked slots read row 0 of x_ubs: any valid address will do when there are no edges
  hipStream_t st = static_cast<hipStream_t>(stream);
#if !defined(K1_STANDALONE)
  if (phases & 256)
    return gatv2_hetero_launch_f32(x_gt, E_seen, seen_off, seen_order, x_ubs, near_off, x_dst, N, seen_params, near_params,
                                   slope, out, ld_out, attn_save_seen, attn_save_near, phases, st);
#endif
  return gatv2_hetero_launch(x_gt, E_seen, seen_off, seen_order, x_ubs, near_off, x_dst, N, seen_params, near_params, slope,
                             out, ld_out, attn_save_seen, attn_save_near, phases, st);
}

extern "C" int uavgnn_gatv2_hetero_fwd(const float* x_gt, int E_seen, const int32_t* seen_off, const int32_t* seen_order,
                                       const float* x_ubs, int E_near, const int32_t* near_off, const float* x_dst, int N,
                                       const float* const* seen_params, const float* const* near_params, int nh, int D_,
                                       float slope, float* out, int ld_out, float* attn_save_seen,
                                       float* attn_save_near, uavgnn_stream_t stream) {
  return uavgnn_gatv2_hetero_fwd_phases(x_gt, E_seen, seen_off, seen_order, x_ubs, E_near, near_off, x_dst, N, seen_params,
                                        near_params, nh, D_, slope, out, ld_out, attn_save_seen, attn_save_near, 3, stream);
}
#endif
