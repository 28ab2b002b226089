// K1: fused GATv2 relation (forward / backward) for the segment layout of the reference's observation graphs.
//
// Replaces dglnn.GATv2Conv.forward as used at /root/reference/algos/madrqn/agents/gnn_agents.py:93-96,:103-104
// (math: SURVEY Appendix A.1/A.4).  Nothing of size [E, H] is ever materialised in HBM:
//   * one wavefront owns one destination node at a time (grid-stride), so every softmax/aggregate is a
//     wave-level segmented reduction;
//   * edge phase: lane <-> edge.  z[u,n] = W_s[n,:] x_u + (b_s + W_d x_v + b_d)[n] is recomputed on the fly from the
//     2..4 input floats held in registers; W_s / attn sit in LDS (broadcast reads), the destination term in a
//     per-wave LDS row;
//   * the aggregate is taken in INPUT space (Appendix A.3 ii): sum_u a_uv el[u] = W_s (sum_u a_uv x_u) + b_s, so only
//     nh*F floats per destination are reduced across lanes;
//   * channel phase: lane <-> output channel, coalesced row stores of out[v, 0..H).
// Backward produces parameter gradients only (observations are leaves), accumulates them in registers per wave
// (lane <-> channel), folds the 4 waves of a workgroup in fixed order through LDS and leaves one partial row per
// workgroup; a second launch sums the rows in fixed order (deterministic, no float atomics).
//
// This file is compiled TWICE (build.py): as itself, and through gatv2_bwd_mfma.hip with UAVGNN_GATV2_BWD_MFMA_TU defined and
// packed fp32 instructions disabled for the whole translation unit.  The second pass emits only the matrix-core instantiation
// of the backward kernel and its launcher; see gatv2_bwd_mfma.hip for why that instantiation may not contain v_pk_*_f32.
#include <cstdlib>

#include "common.h"
#include "k1_x3.h"

namespace uavgnn {
// defined in gatv2_bwd_mfma.hip (the other pass over this file): launches gatv2_bwd_kernel<4, 4, 64, true, true>, cls = 1
int gatv2_bwd_mfma_launch(const float* x_src, const float* x_dst, const int32_t* seg_off, const int32_t* dst_order, int N,
                          const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn, float slope,
                          const float* out, const float* d_out, int ld_out, const float* a_save, float* partial,
                          int onepass_max_deg, int grid, hipStream_t st);
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;
#ifndef UAVGNN_BWD_UNROLL
#define UAVGNN_BWD_UNROLL 4
#endif
#ifndef UAVGNN_BWD_OCC
#define UAVGNN_BWD_OCC 2
#endif

template <int FS>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&x)[FS]);
template <>
__device__ __forceinline__ void load_row<4>(const float* __restrict__ p, float (&x)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
}
template <>
__device__ __forceinline__ void load_row<2>(const float* __restrict__ p, float (&x)[2]) {
  const float2 t = *reinterpret_cast<const float2*>(p);
  x[0] = t.x; x[1] = t.y;
}

// ---------------------------------------------------------------------------------------------------------------
template <int FS, int NH, int D>
__global__ __launch_bounds__(kThreads) void gatv2_fwd_kernel(
    const float* __restrict__ x_src, const float* __restrict__ x_dst, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ dst_order, int N,
    const float* __restrict__ W_s, const float* __restrict__ b_s, const float* __restrict__ W_d,
    const float* __restrict__ b_d, const float* __restrict__ attn, const float* __restrict__ W_r,
    const float* __restrict__ b_r, float slope, float* __restrict__ out, int ld_out, float* __restrict__ a_save) {
  constexpr int H = NH * D;
  constexpr int J = (H + kWave - 1) / kWave;
  __shared__ float sW[H * FS];
  __shared__ float sAttn[H];
  __shared__ float sC[kWavesPerBlock][H];
  __shared__ float sS[kWavesPerBlock][NH * FS];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int i = tid; i < H * FS; i += kThreads) sW[i] = W_s[i];
  for (int i = tid; i < H; i += kThreads) sAttn[i] = attn[i];

  float wd0[J], wd1[J], bc[J], wr0[J], wr1[J], br[J], bs[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int n = lane + kWave * j;
    const bool ok = n < H;
    wd0[j] = ok ? W_d[n * 2 + 0] : 0.f;
    wd1[j] = ok ? W_d[n * 2 + 1] : 0.f;
    bs[j] = ok ? b_s[n] : 0.f;
    bc[j] = ok ? b_d[n] + bs[j] : 0.f;
    wr0[j] = ok ? W_r[n * 2 + 0] : 0.f;
    wr1[j] = ok ? W_r[n * 2 + 1] : 0.f;
    br[j] = (ok && b_r != nullptr) ? b_r[n] : 0.f;
  }
  __syncthreads();

  float* __restrict__ cw = sC[wave];
  float* __restrict__ sw = sS[wave];

  for (int it = blockIdx.x * kWavesPerBlock + wave; it < N; it += gridDim.x * kWavesPerBlock) {
    const int v = dst_order ? dst_order[it] : it;
    const float xv0 = x_dst[2 * v], xv1 = x_dst[2 * v + 1];
    const int e0 = seg_off[v];
    const int deg = seg_off[v + 1] - e0;

    float res[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = lane + kWave * j;
      res[j] = fmaf(wr1[j], xv1, fmaf(wr0[j], xv0, br[j]));
      if (n < H) cw[n] = fmaf(wd1[j], xv1, fmaf(wd0[j], xv0, bc[j]));
    }
    float* __restrict__ orow = out + static_cast<size_t>(v) * ld_out;
    if (deg == 0) {  // allow_zero_in_degree: aggregate is 0, only residual + ReLU (KAT 1)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int n = lane + kWave * j;
        if (n < H) orow[n] = fmaxf(res[j], 0.f);
      }
      continue;
    }
    wave_sync();

    // ---- edge phase: lane-local online softmax state per head -----------------------------------------------
    float m[NH], den[NH], s[NH][FS];
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      m[k] = -INFINITY;
      den[k] = 0.f;
#pragma unroll
      for (int f = 0; f < FS; ++f) s[k][f] = 0.f;
    }
    for (int base = 0; base < deg; base += kWave) {
      const bool valid = base + lane < deg;
      const int u = e0 + base + lane;
      float x[FS];
      if (valid) {
        load_row<FS>(x_src + static_cast<size_t>(u) * FS, x);
      } else {
#pragma unroll
        for (int f = 0; f < FS; ++f) x[f] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
          const int n = k * D + d;
          float z = cw[n];
#pragma unroll
          for (int f = 0; f < FS; ++f) z = fmaf(sW[n * FS + f], x[f], z);
          const float lz = z > 0.f ? z : slope * z;
          acc = fmaf(sAttn[n], lz, acc);
        }
        if (valid) {
          if (a_save != nullptr) a_save[static_cast<size_t>(u) * NH + k] = acc;  // raw score, normalised below
          const float mn = fmaxf(m[k], acc);
          const float sc = expf(m[k] - mn);  // exp(-inf) = 0 on the first edge
          const float p = expf(acc - mn);
          den[k] = fmaf(den[k], sc, p);
#pragma unroll
          for (int f = 0; f < FS; ++f) s[k][f] = fmaf(s[k][f], sc, p * x[f]);
          m[k] = mn;
        }
      }
    }
    // ---- combine lanes: segment softmax + input-space aggregate --------------------------------------------
    float mx[NH], inv[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      mx[k] = wave_max(m[k]);
      const float sc = (m[k] == -INFINITY) ? 0.f : expf(m[k] - mx[k]);
      const float dn = wave_sum(den[k] * sc);
      inv[k] = 1.f / dn;
#pragma unroll
      for (int f = 0; f < FS; ++f) {
        const float t = wave_sum(s[k][f] * sc);
        if (lane == 0) sw[k * FS + f] = t * inv[k];
      }
    }
    if (a_save != nullptr) {
      for (int base = 0; base < deg; base += kWave) {
        if (base + lane < deg) {
          float* ap = a_save + static_cast<size_t>(e0 + base + lane) * NH;
#pragma unroll
          for (int k = 0; k < NH; ++k) ap[k] = expf(ap[k] - mx[k]) * inv[k];
        }
      }
    }
    wave_sync();
    // ---- channel phase: project the aggregate once per destination, add residual, ReLU -----------------------
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = lane + kWave * j;
      if (n < H) {
        const int k = n / D;
        float agg = bs[j];
#pragma unroll
        for (int f = 0; f < FS; ++f) agg = fmaf(sW[n * FS + f], sw[k * FS + f], agg);
        orow[n] = fmaxf(agg + res[j], 0.f);
      }
    }
    wave_sync();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Partial-gradient row layout (floats): dW_s[H*FS] | db_s[H] | dW_d[2H] | db_d[H] | dattn[H] | dW_r[2H] | db_r[H]
constexpr int kRowRorCtl = 0x120, kQuadXor1Ctl = 0xB1, kQuadXor2Ctl = 0x4E, kHalfMirrorCtl = 0x141;
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_f<kRowRorCtl + 8>(v);
  v += dpp_f<kRowRorCtl + 4>(v);
  v += dpp_f<kRowRorCtl + 2>(v);
  v += dpp_f<kRowRorCtl + 1>(v);
  return v;
}
__device__ __forceinline__ float half8_allsum(float v) {
  v += dpp_f<kQuadXor1Ctl>(v);
  v += dpp_f<kQuadXor2Ctl>(v);
  v += dpp_f<kHalfMirrorCtl>(v);
  return v;
}
// total over the 64 lanes, the same value in every lane, without the LDS crossbar (__shfl_xor = ds_bpermute_b32: six dependent
// round trips of ~100 cycles each): row sums by four DPP adds, rows 1 and 3 take lane 15 / 47 of the row below (row_bcast:15),
// rows 2 and 3 lane 31 (row_bcast:31) - the last row then holds ((r2 + r3) + (r0 + r1)) - and lane 63 is read back through an
// SGPR.  Seven instructions per value; deterministic (a fixed tree).
__device__ __forceinline__ float wave_total_dpp(float v) {
  v = row16_allsum(v);
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x142, 0xa, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x143, 0xc, 0xf, false));
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

template <int FS>
__host__ __device__ constexpr int partial_len(int H) { return H * (FS + 8); }

// MF (F_src = 4, nh = 4, D = 64 only; launched with cls = 1): the per-(edge, channel) sums S1, S2 of destinations with kMfMinDeg ..
// onepass_max_deg in-edges on the bf16 matrix cores (run_edges_mfma below) instead of the packed-FMA loop of run_edges.
typedef float bwd_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMfMinDeg = 16;

template <int FS, int NH, int D, bool CHUNK, bool MF = false>
__global__ __launch_bounds__(kThreads, UAVGNN_BWD_OCC) void gatv2_bwd_kernel(
    const float* __restrict__ x_src, const float* __restrict__ x_dst, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ dst_order, int N,
    const float* __restrict__ W_s, const float* __restrict__ b_s, const float* __restrict__ W_d,
    const float* __restrict__ b_d, const float* __restrict__ attn, float slope, const float* __restrict__ out,
    const float* __restrict__ d_out, int ld_out, const float* __restrict__ a_save, float* __restrict__ partial,
    int onepass_max_deg, int cls) {
  // Per destination v (one wavefront), head k, channel n = (k,d), in-edges u with attention a_uk:
  //   g = d_out * [out > 0];  G[k] = sum_d g[n] W_s[n,:];  de_uk = a_uk (G[k].x_u - T[k]),  T[k] = sum_u a_uk G[k].x_u
  // With lrelu'(z) = c_lin + c_abs sgn(z), c_lin = (1+s)/2, c_abs = (1-s)/2 (SURVEY A.3 i) everything that is linear in
  // the edge collapses to per-(destination, head) sums in INPUT space, and only the sign part is per (edge, channel):
  //   s_un = +de_uk if z_un > 0 else -de_uk;   S1[n] = sum_u s_un;   S2[n,:] = sum_u s_un x_u
  //   P[k,:] = sum_u de_uk x_u;   Sb[k,:] = sum_u a_uk x_u          (sum_u de_uk = 0)
  //   d attn[n]  += c_abs (W_s[n,:].S2[n,:] + c[n] S1[n]) + c_lin W_s[n,:].P[k,:]
  //   d W_s[n,:] += attn[n] (c_abs S2[n,:] + c_lin P[k,:]) + g[n] Sb[k,:]
  //   d er[v,n]   = attn[n] c_abs S1[n]  ->  d b_s += g + der, d b_d += der, d W_d += der (x) x_v
  // i.e. F_src FMAs (z) + 2 (sign select) + 1 + F_src FMAs per (edge, channel) instead of ~2 F_src + 8.
  constexpr int H = NH * D;
  constexpr int J = (H + kWave - 1) / kWave;
  constexpr int KF = NH * FS;
  static_assert(KF <= kWave && (kWave % KF) == 0, "nh*F_src must divide 64");
  constexpr int PARTS = kWave / KF;
  constexpr int P = partial_len<FS>(H);
  constexpr int ES = FS + 2 * NH;  // staged floats per edge: x[FS], de[NH], a[NH]

  // LDS, carved by hand from one array
  static_assert(!MF || (FS == 4 && NH == 4 && D == 64), "matrix-core backward: F_src = 4, nh = 4, D = 64");
  constexpr int szG = kWavesPerBlock * H * 4, szGk = kWavesPerBlock * KF * 4, szPS = kWavesPerBlock * 2 * KF * 4;
  constexpr int szE = kWavesPerBlock * 2 * kWave * ES * 4;      // up to 128 staged edges per wavefront (two 64-edge chunks)
  constexpr int szV = MF ? kWavesPerBlock * NH * kWave * 16 : 0;   // MF: A operands of the accumulation products of one pair of edge tiles, [head][lane], 4 KB per wavefront
  constexpr int szW = H * FS * 4, szRed = P * 4;
  constexpr int szWop = MF ? (H / 16) * kWave * 16 : 0;         // MF: B operands of the score products (rows of W_s as exact bf16 triples, [channel tile][lane]: 16 KB)
  constexpr int oG = 0, oGk = oG + szG, oPS = oGk + szGk, oE = (oPS + szPS + 15) & ~15, oV = oE + szE, oW = oV + szV;
  constexpr int oWop = oW + szW;
  constexpr int oRed = MF ? oWop : oWop + szWop;   // MF: the fold buffer of the last phase takes the place of the score operands (barrier in between)
  static_assert(!MF || szRed <= szWop, "fold buffer aliases the score operands");
  // MF: the destination term -c of the score as the bias words of the W operands: per wavefront [channel tile][K group 0, 1][16
  // channels] dwords (2 KB), and one all-zero table of the same shape for the lanes of K groups 2 and 3 (2 KB per workgroup)
  constexpr int szC = MF ? (kWavesPerBlock + 1) * (H / 16) * 32 * 4 : 0;
  constexpr int oC = oWop + szWop;
  constexpr int kLdsBytes = MF ? oC + szC : oRed + szRed;
  static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
  float* const sW = reinterpret_cast<float*>(lds + oW);
  float* const sRed = reinterpret_cast<float*>(lds + oRed);
  k1_u32x4* const sWop = reinterpret_cast<k1_u32x4*>(lds + oWop);

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float c_lin = 0.5f * (1.f + slope), c_abs = 0.5f * (1.f - slope);

  for (int i = tid; i < H * FS; i += kThreads) sW[i] = W_s[i];
  if constexpr (MF) {
    for (int i = tid; i < (H / 16) * kWave; i += kThreads) {   // lane (j = channel of the tile, kg = feature)
      const int ct = i >> 6, l = i & 63;
      sWop[i] = k1_a_operand(0.f - W_s[(16 * ct + (l & 15)) * FS + (l >> 4)], 0.f, 0);   // -W: the score MFMA yields -(z + c)
    }
    for (int i = tid; i < (H / 16) * 32; i += kThreads) reinterpret_cast<unsigned*>(lds + oC)[kWavesPerBlock * (H / 16) * 32 + i] = 0u;
    for (int i = lane; i < NH * kWave; i += kWave)   // the zero row of every operand is never written again
      (reinterpret_cast<k1_u32x4*>(lds + oV) + (tid >> 6) * NH * kWave)[i] = k1_u32x4{0u, 0u, 0u, 0u};
  }

  float Ws[J][FS], att[J], wd0[J], wd1[J], bc[J];
  int kj[J];
  float aWs[J][FS], abs_[J], aWd0[J], aWd1[J], abd[J], aatt[J], aWr0[J], aWr1[J], abr[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int n = lane + kWave * j;
    const bool ok = n < H;
    kj[j] = ok ? n / D : 0;
#pragma unroll
    for (int f = 0; f < FS; ++f) {
      Ws[j][f] = ok ? W_s[n * FS + f] : 0.f;
      aWs[j][f] = 0.f;
    }
    att[j] = ok ? attn[n] : 0.f;
    wd0[j] = ok ? W_d[n * 2 + 0] : 0.f;
    wd1[j] = ok ? W_d[n * 2 + 1] : 0.f;
    bc[j] = ok ? b_d[n] + b_s[n] : 0.f;
    abs_[j] = aWd0[j] = aWd1[j] = abd[j] = aatt[j] = aWr0[j] = aWr1[j] = abr[j] = 0.f;
  }
  // lanes 0..KF-1 additionally own one (head, feature) pair of the input-space sums P and Sb
  const int pk = (lane < KF) ? lane / FS : 0, pf = (lane < KF) ? lane % FS : 0;
  __syncthreads();

  float* __restrict__ gw = reinterpret_cast<float*>(lds + oG) + wave * H;
  float* __restrict__ gk = reinterpret_cast<float*>(lds + oGk) + wave * KF;
  float* __restrict__ ps = reinterpret_cast<float*>(lds + oPS) + wave * 2 * KF;
  float* __restrict__ ew = reinterpret_cast<float*>(lds + oE) + wave * 2 * kWave * ES;
  k1_u32x4* __restrict__ vw_base = reinterpret_cast<k1_u32x4*>(lds + oV) + (MF ? wave * NH * kWave : 0);

  // rows of the forward output and of its gradient for destination v (lane <-> channel)
  auto load_rows = [&](const int v, float (&o)[J], float (&gr)[J]) {
    const float* __restrict__ orow = out + static_cast<size_t>(v) * ld_out;
    const float* __restrict__ grow = d_out + static_cast<size_t>(v) * ld_out;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = lane + kWave * j;
      o[j] = n < H ? orow[n] : 0.f;
      gr[j] = n < H ? grow[n] : 0.f;
    }
  };

  auto process = [&](const float (&o_)[J], const float (&gr_)[J], const int e0, const int deg, const float xv0,
                     const float xv1) {
    float g[J], c[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      g[j] = o_[j] > 0.f ? gr_[j] : 0.f;  // ReLU mask
      aWr0[j] = fmaf(g[j], xv0, aWr0[j]);
      aWr1[j] = fmaf(g[j], xv1, aWr1[j]);
      abr[j] += g[j];
      c[j] = fmaf(wd1[j], xv1, fmaf(wd0[j], xv0, bc[j]));
    }
    if (deg == 0) return;

    // G[k][f] = sum_d g[k,d] W_s[k,d,f]
    if constexpr (D == kWave) {   // head k = register j: NH * FS wave totals straight from the registers, no LDS round trip
      float gkf = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int f = 0; f < FS; ++f) {
          const float tot = wave_total_dpp(g[j] * Ws[j][f]);
          gkf = (lane == j * FS + f) ? tot : gkf;
        }
      if (lane < KF) gk[lane] = gkf;
    } else {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = lane + kWave * j;
      if (n < H) gw[n] = g[j];
    }
    wave_sync();
    {
      const int kf = lane / PARTS, part = lane % PARTS;
      const int k = kf / FS, f = kf % FS;
      float gp = 0.f;
      for (int d = part; d < D; d += PARTS) {
        const int n = k * D + d;
        gp = fmaf(gw[n], sW[n * FS + f], gp);
      }
#pragma unroll
      for (int o = PARTS / 2; o > 0; o >>= 1) gp += __shfl_xor(gp, o);
      if (part == 0) gk[kf] = gp;
    }
    }
    wave_sync();
    float S1[J], S2[J][FS];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      S1[j] = 0.f;
#pragma unroll
      for (int f = 0; f < FS; ++f) S2[j][f] = 0.f;
    }
    float accP = 0.f, accSb = 0.f;   // lanes < KF: P[pk][pf], Sb[pk][pf]
    // the per-(edge, channel) sums over `cnt` staged edges starting at staged slot `s0` (lane <-> channel, edge data is a
    // broadcast LDS read)
    auto run_edges = [&](const int s0, const int cnt) {
#pragma unroll UAVGNN_BWD_UNROLL
      for (int ii = 0; ii < cnt; ++ii) {
        const int i = s0 + ii;
        float xe[FS];
#pragma unroll
        for (int f = 0; f < FS; ++f) xe[f] = ew[i * ES + f];
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const float dek = ew[i * ES + FS + kj[j]];
          float z = c[j];
#pragma unroll
          for (int f = 0; f < FS; ++f) z = fmaf(Ws[j][f], xe[f], z);
          const float sde = z > 0.f ? dek : -dek;
          S1[j] += sde;
#pragma unroll
          for (int f = 0; f < FS; ++f) S2[j][f] = fmaf(sde, xe[f], S2[j][f]);
        }
        const float xpf = ew[i * ES + pf];
        accP = fmaf(ew[i * ES + FS + pk], xpf, accP);
        accSb = fmaf(ew[i * ES + FS + NH + pk], xpf, accSb);
      }
    };
    // ---- the same sums on the bf16 matrix cores (MF; staged edges 0 .. deg - 1, deg <= 128) --------------------------------------
    auto run_edges_mfma = [&](const int deg_) {
      if constexpr (MF) {
        constexpr int CT = H / 16;
        bwd_f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = bwd_f32x4{0.f, 0.f, 0.f, 0.f};
        const bwd_f32x4 czero = {0.f, 0.f, 0.f, 0.f};
        unsigned k_sign = 0x80008000u, k_one = 0x3F803F80u;
        asm volatile("" : "+v"(k_sign), "+v"(k_one));
        const int j16 = lane & 15, g4 = lane >> 4;
        k1_u32x4* __restrict__ vw = vw_base;
        {
          const int kf = lane & 15, part = lane >> 4, k = kf >> 2, f = kf & 3;
          float p = 0.f, sb = 0.f;
          const int e_end = (deg_ + 15) & ~15;
#pragma unroll 2
          for (int e = part; e < e_end; e += 16) {
            float x[4], d4[4], a4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              x[u] = ew[(e + 4 * u) * ES + f];
              d4[u] = ew[(e + 4 * u) * ES + FS + k];
              a4[u] = ew[(e + 4 * u) * ES + FS + NH + k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              p = fmaf(d4[u], x[u], p);
              sb = fmaf(a4[u], x[u], sb);
            }
          }
          p += __shfl_xor(p, 16);
          sb += __shfl_xor(sb, 16);
          p += __shfl_xor(p, 32);
          sb += __shfl_xor(sb, 32);
          accP = p;
          accSb = sb;
        }
        for (int tp = 0; 32 * tp < deg_; ++tp) {
          {
            const int ep = lane & 15, h = lane >> 4;
            const int e_a = 2 * ep, i16 = e_a & 15;
            const float* er = ew + (32 * tp + e_a) * ES;
            const bwd_f32x4 xa = *reinterpret_cast<const bwd_f32x4*>(er), xb = *reinterpret_cast<const bwd_f32x4*>(er + ES);
            const float da = er[FS + h], db = er[ES + FS + h];
            const float va[5] = {da, da * xa[0], da * xa[1], da * xa[2], da * xa[3]};
            const float vb[5] = {db, db * xb[0], db * xb[1], db * xb[2], db * xb[3]};
            const int skew = (i16 >> 2) + 4 * h;
            unsigned* dst = reinterpret_cast<unsigned*>(vw) + (h * kWave + (i16 >> 2) * 16) * 4 + (((i16 & 3) + 4 * (e_a >> 4)) >> 1);
#pragma unroll
            for (int f = 0; f < 5; ++f) {
              const K1Split sa = k1_split(va[f]), sb2 = k1_split(vb[f]);
              dst[((0 * 5 + f + skew) & 15) * 4] = (sa.h1 & 0xffffu) | (sb2.h1 & 0xffff0000u);
              dst[((1 * 5 + f + skew) & 15) * 4] = (sa.h2 & 0xffffu) | (sb2.h2 & 0xffff0000u);
              dst[((2 * 5 + f + skew) & 15) * 4] = (sa.h3 & 0xffffu) | (sb2.h3 & 0xffff0000u);
            }
          }
          k1_bf16x8 xop[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) xop[t] = k1_b_operand(ew[(32 * tp + 16 * t + j16) * ES + g4], 0x3F803F80u);
          wave_sync_lds();
          k1_u32x4 vop = vw[16 * g4 + ((j16 + g4) & 15)], vop_n = vw[kWave + 16 * g4 + ((j16 + g4 + 4) & 15)];
          // lane (channel j16, K group g4): W planes of feature g4 from the shared image, bias word from the wavefront's table
          const unsigned* __restrict__ cwl = reinterpret_cast<const unsigned*>(lds + oC) +
              (g4 < 2 ? wave * (H / 16) * 32 + 16 * g4 + j16 : kWavesPerBlock * (H / 16) * 32 + (lane & 31));
          auto w_operand = [&](const int ct_) {
            k1_u32x4 w = sWop[ct_ * kWave + lane];
            w[3] = cwl[ct_ * 32];
            return __builtin_bit_cast(k1_bf16x8, w);
          };
          k1_bf16x8 w_nn = w_operand(1);
          bwd_f32x4 d0, d1;
          {
            const k1_bf16x8 w0 = w_operand(0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xop[0], w0, czero, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xop[1], w0, czero, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            bwd_f32x4 e0 = d0, e1 = d1;
            if (ct + 1 < CT) {
              e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xop[0], w_nn, czero, 0, 0, 0);
              e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xop[1], w_nn, czero, 0, 0, 0);
              if (ct + 2 < CT) w_nn = w_operand(ct + 2);
              __builtin_amdgcn_sched_barrier(0);
            }
            k1_u32x4 sg;
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // d = -(z + c): the sign bits of two edges' values side by side over 1.0
              const unsigned p0 = __builtin_amdgcn_perm(__float_as_uint(d0[2 * q + 1]), __float_as_uint(d0[2 * q]), 0x07060302u);
              const unsigned p1 = __builtin_amdgcn_perm(__float_as_uint(d1[2 * q + 1]), __float_as_uint(d1[2 * q]), 0x07060302u);
              sg[q] = (p0 & k_sign) | (k_one & ~k_sign);
              sg[2 + q] = (p1 & k_sign) | (k_one & ~k_sign);
            }
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k1_bf16x8, vop), __builtin_bit_cast(k1_bf16x8, sg),
                                                              acc[ct], 0, 0, 0);
            if ((ct & 3) == 3 && ct + 1 < CT) {
              vop = vop_n;
              if (ct + 5 < CT) vop_n = vw[((ct + 5) >> 2) * kWave + 16 * g4 + ((j16 + g4 + 4 * ((ct + 5) >> 2)) & 15)];
            }
            __builtin_amdgcn_sched_barrier(0);
            d0 = e0;
            d1 = e1;
          }
          wave_sync_lds();
        }
        constexpr int kCvLd = 20;
        float* __restrict__ cv = ew;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) *reinterpret_cast<bwd_f32x4*>(cv + (c4 * 16 + j16) * kCvLd + 4 * g4) = acc[4 * k + c4];
          wave_sync_lds();
          float r[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bwd_f32x4 t4 = *reinterpret_cast<const bwd_f32x4*>(cv + lane * kCvLd + 4 * q);
            r[4 * q] = t4[0]; r[4 * q + 1] = t4[1]; r[4 * q + 2] = t4[2]; r[4 * q + 3] = t4[3];
          }
          S1[k] = -((r[0] + r[5]) + r[10]);
#pragma unroll
          for (int f = 0; f < FS; ++f) S2[k][f] = -((r[1 + f] + r[6 + f]) + r[11 + f]);
          wave_sync_lds();
        }
      }
    };
    // T[k] = sum_u a_uk (G[k].x_u): one wave reduction per head
    float T[NH];
    if (deg <= onepass_max_deg) {
      // Up to 128 in-edges (every BASELINE configuration): ONE pass over the edge data.  x_u, a_uk and G[k].x_u of all edges
      // are staged in LDS while T accumulates (same summation order as the two-pass path: chunk 0, then chunk 1, then the
      // wave reduction - bit-identical), then every lane turns the staged dots of its edges into de_uk = a_uk (dot - T[k]) in
      // place.  The second pass of the general path below re-loads x_u and a_uk and recomputes the dots.
      float t[NH];
#pragma unroll
      for (int k = 0; k < NH; ++k) t[k] = 0.f;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const int slot = ch * kWave + lane;
        if (ch * kWave < deg) {
          float x[FS], a[NH], dot[NH];
          if (slot < deg) {
            const int u = e0 + slot;
            load_row<FS>(x_src + static_cast<size_t>(u) * FS, x);
#pragma unroll
            for (int k = 0; k < NH; ++k) {
              float d = 0.f;
#pragma unroll
              for (int f = 0; f < FS; ++f) d = fmaf(gk[k * FS + f], x[f], d);
              dot[k] = d;
              a[k] = a_save[static_cast<size_t>(u) * NH + k];
              t[k] = fmaf(a[k], d, t[k]);
            }
          } else {
#pragma unroll
            for (int f = 0; f < FS; ++f) x[f] = 0.f;
#pragma unroll
            for (int k = 0; k < NH; ++k) dot[k] = a[k] = 0.f;
          }
#pragma unroll
          for (int f = 0; f < FS; ++f) ew[slot * ES + f] = x[f];
#pragma unroll
          for (int k = 0; k < NH; ++k) {
            ew[slot * ES + FS + k] = dot[k];
            ew[slot * ES + FS + NH + k] = a[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NH; ++k) T[k] = wave_total_dpp(t[k]);
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const int slot = ch * kWave + lane;
        if (ch * kWave < deg) {
#pragma unroll
          for (int k = 0; k < NH; ++k)      // the lane's own slots: written above by this lane, no synchronisation needed
            ew[slot * ES + FS + k] = ew[slot * ES + FS + NH + k] * (ew[slot * ES + FS + k] - T[k]);
        }
      }
      if constexpr (MF) {
        {
          unsigned* __restrict__ cwt = reinterpret_cast<unsigned*>(lds + oC) + wave * (H / 16) * 32;
#pragma unroll
          for (int j = 0; j < J; ++j) {   // channel lane + 64 j = tile (lane >> 4) + 4 j, column lane & 15
            const K1Split sc = k1_split(0.f - c[j]);
            unsigned* q = cwt + ((lane >> 4) + 4 * j) * 32 + (lane & 15);
            q[0] = (sc.h1 & 0xffffu) | (sc.h2 & 0xffff0000u);
            q[16] = sc.h3 & 0xffffu;
          }
        }
        wave_sync();
        run_edges_mfma(deg);
      } else {
        wave_sync();
        run_edges(0, deg);
      }
      wave_sync();
    } else if constexpr (!MF) {
      {
        float t[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) t[k] = 0.f;
        for (int base = 0; base < deg; base += kWave) {
          if (base + lane < deg) {
            const int u = e0 + base + lane;
            float x[FS];
            load_row<FS>(x_src + static_cast<size_t>(u) * FS, x);
#pragma unroll
            for (int k = 0; k < NH; ++k) {
              float dot = 0.f;
#pragma unroll
              for (int f = 0; f < FS; ++f) dot = fmaf(gk[k * FS + f], x[f], dot);
              t[k] = fmaf(a_save[static_cast<size_t>(u) * NH + k], dot, t[k]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) T[k] = wave_total_dpp(t[k]);
      }
      for (int base = 0; base < deg; base += kWave) {
        {  // stage x_u, de_uk, a_uk of up to 64 edges in LDS (lane <-> edge)
          const bool valid = base + lane < deg;
          const int u = e0 + base + lane;
          float x[FS], de[NH], a[NH];
          if (valid) {
            load_row<FS>(x_src + static_cast<size_t>(u) * FS, x);
#pragma unroll
            for (int k = 0; k < NH; ++k) {
              float dot = 0.f;
#pragma unroll
              for (int f = 0; f < FS; ++f) dot = fmaf(gk[k * FS + f], x[f], dot);
              a[k] = a_save[static_cast<size_t>(u) * NH + k];
              de[k] = a[k] * (dot - T[k]);
            }
          } else {
#pragma unroll
            for (int f = 0; f < FS; ++f) x[f] = 0.f;
#pragma unroll
            for (int k = 0; k < NH; ++k) de[k] = a[k] = 0.f;
          }
#pragma unroll
          for (int f = 0; f < FS; ++f) ew[lane * ES + f] = x[f];
#pragma unroll
          for (int k = 0; k < NH; ++k) {
            ew[lane * ES + FS + k] = de[k];
            ew[lane * ES + FS + NH + k] = a[k];
          }
        }
        wave_sync();
        run_edges(0, min(kWave, deg - base));
        wave_sync();
      }
    }
    if (lane < KF) {
      ps[lane] = accP;
      ps[KF + lane] = accSb;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int k = kj[j];
      float wS2 = 0.f, wP = 0.f;
#pragma unroll
      for (int f = 0; f < FS; ++f) {
        const float pkf = ps[k * FS + f], sbf = ps[KF + k * FS + f];
        wS2 = fmaf(Ws[j][f], S2[j][f], wS2);
        wP = fmaf(Ws[j][f], pkf, wP);
        aWs[j][f] += att[j] * fmaf(c_abs, S2[j][f], c_lin * pkf) + g[j] * sbf;
      }
      aatt[j] += fmaf(c_abs, fmaf(c[j], S1[j], wS2), c_lin * wP);
      const float der = att[j] * c_abs * S1[j];
      abs_[j] += g[j] + der;
      abd[j] += der;
      aWd0[j] = fmaf(der, xv0, aWd0[j]);
      aWd1[j] = fmaf(der, xv1, aWd1[j]);
    }
    wave_sync();
  };

  const int stride = gridDim.x * kWavesPerBlock;
  const int it0 = blockIdx.x * kWavesPerBlock + wave;
  if constexpr (CHUNK) {   // sparse batches: 64 destinations' meta data by vector loads + v_readlane (see the forward)
    for (int kb = 0; it0 + kb * stride < N; kb += kWave) {
      const int my_it = it0 + (kb + lane) * stride;
      const bool mine = my_it < N;
      const int m_v = mine ? (dst_order ? dst_order[my_it] : my_it) : 0;
      const int m_e0 = mine ? seg_off[m_v] : 0;
      const int m_e1 = mine ? seg_off[m_v + 1] : 0;
      const float2 m_xv = mine ? *reinterpret_cast<const float2*>(x_dst + 2 * m_v) : make_float2(0.f, 0.f);
      const int cnt = min(kWave, (N - it0 - kb * stride + stride - 1) / stride);
      auto in_cls = [&](const int i) {
        const int dg = __builtin_amdgcn_readlane(m_e1, i) - __builtin_amdgcn_readlane(m_e0, i);
        const bool mf = dg >= kMfMinDeg && dg <= onepass_max_deg;
        return cls == 0 || (mf == (cls == 1));
      };
      float on[J], gn[J];                       // rows of the NEXT destination are in flight while this one computes
      int ii = 0;
      while (ii < cnt && !in_cls(ii)) ++ii;
      if (ii < cnt) load_rows(__builtin_amdgcn_readlane(m_v, ii), on, gn);
      while (ii < cnt) {
        float oc[J], gc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
          oc[j] = on[j];
          gc[j] = gn[j];
        }
        int nx = ii + 1;
        while (nx < cnt && !in_cls(nx)) ++nx;
        if (nx < cnt) load_rows(__builtin_amdgcn_readlane(m_v, nx), on, gn);
        const int e0 = __builtin_amdgcn_readlane(m_e0, ii);
        process(oc, gc, e0, __builtin_amdgcn_readlane(m_e1, ii) - e0,
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xv.x), ii)),
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xv.y), ii)));
        ii = nx;
      }
    }
  } else {
    if (it0 < N) {
      float on[J], gn[J];
      int vn = dst_order ? dst_order[it0] : it0;
      load_rows(vn, on, gn);
      for (int it = it0; it < N; it += stride) {
        const int v = vn;
        float oc[J], gc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
          oc[j] = on[j];
          gc[j] = gn[j];
        }
        if (it + stride < N) {   // rows of the next destination are in flight while this one computes
          vn = dst_order ? dst_order[it + stride] : it + stride;
          load_rows(vn, on, gn);
        }
        const int e0 = seg_off[v];
        process(oc, gc, e0, seg_off[v + 1] - e0, x_dst[2 * v], x_dst[2 * v + 1]);
      }
    }
  }

  // fold the 4 waves in fixed order through LDS, then one partial row per workgroup
  if constexpr (MF) __syncthreads();   // the fold buffer aliases the score operands
  for (int w = 0; w < kWavesPerBlock; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int n = lane + kWave * j;
        if (n < H) {
          auto put = [&](int idx, float val) { sRed[idx] = (w == 0) ? val : sRed[idx] + val; };
#pragma unroll
          for (int f = 0; f < FS; ++f) put(n * FS + f, aWs[j][f]);
          int o = H * FS;
          put(o + n, abs_[j]);
          o += H;
          put(o + 2 * n, aWd0[j]);
          put(o + 2 * n + 1, aWd1[j]);
          o += 2 * H;
          put(o + n, abd[j]);
          o += H;
          put(o + n, aatt[j]);
          o += H;
          put(o + 2 * n, aWr0[j]);
          put(o + 2 * n + 1, aWr1[j]);
          o += 2 * H;
          put(o + n, abr[j]);
        }
      }
    }
    __syncthreads();
  }
  float* __restrict__ prow = partial + static_cast<size_t>(blockIdx.x) * P;
  for (int i = tid; i < P; i += kThreads) prow[i] = sRed[i];
}

#ifdef UAVGNN_GATV2_BWD_MFMA_TU
}  // namespace

int gatv2_bwd_mfma_launch(const float* x_src, const float* x_dst, const int32_t* seg_off, const int32_t* dst_order, int N,
                          const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn, float slope,
                          const float* out, const float* d_out, int ld_out, const float* a_save, float* partial,
                          int onepass_max_deg, int grid, hipStream_t st) {
  hipLaunchKernelGGL((gatv2_bwd_kernel<4, 4, 64, true, true>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off, dst_order,
                     N, W_s, b_s, W_d, b_d, attn, slope, out, d_out, ld_out, a_save, partial, onepass_max_deg, 1);
  return launch_status();
}
}  // namespace uavgnn
#else   // everything below belongs to the first pass only

// Sums `rows` partial rows of length P in a fixed order: block = 64 columns x 16 row-groups.
struct GradPtrs {
  float* p[7];
  int off[8];
};

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, int rows, int P,
                                                               GradPtrs gp) {
  __shared__ float sm[16][kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int col = blockIdx.x * kWave + lane;
  float acc = 0.f;
  if (col < P) {
    int r = wave;
    for (; r + 48 < rows; r += 64) {
      const float a0 = partial[static_cast<size_t>(r) * P + col];
      const float a1 = partial[static_cast<size_t>(r + 16) * P + col];
      const float a2 = partial[static_cast<size_t>(r + 32) * P + col];
      const float a3 = partial[static_cast<size_t>(r + 48) * P + col];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; r < rows; r += 16) acc += partial[static_cast<size_t>(r) * P + col];
  }
  sm[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col < P) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sm[w][lane];
    int seg = 0;
#pragma unroll
    for (int i = 1; i < 7; ++i) seg += (col >= gp.off[i]) ? 1 : 0;
    gp.p[seg][col - gp.off[seg]] = t;
  }
}

constexpr int kMaxBwdBlocks = 1024;

inline int bwd_blocks(int N) { return capped_grid(N, kWavesPerBlock, kMaxBwdBlocks); }


// ---------------------------------------------------------------------------------------------------------------
// K1 backward for LOW-DEGREE two-feature relations (`near`: n - 1 <= 8 in-edges of 2 features; nh = 4, D = 64).
// The generic kernel above owns one destination per wavefront and pays ~1400 cycles of per-destination bookkeeping
// (LDS round trips for G, four 64-lane reductions for T, edge staging for up to 64 edges) around ~700 cycles of
// per-(edge, channel) work when the degree is 7.  Here a wavefront owns a PAIR of destinations (2p, 2p+1):
//   * lane <-> four consecutive channels 4*lane..4*lane+3 of head k = lane>>4 for everything per channel (rows of `out`
//     / `d_out` are one 16-byte load per lane), and lane = (head k, destination half, edge slot) for everything per edge
//     - the same 16-lane row serves head k in both roles, so G[k], T[k], P[k], Sb[k] never leave their row: the
//     reductions are DPP all-reduces over 16 (channels) or 8 (edge slots of one destination) lanes, no LDS, no barrier;
//   * only the per-edge (x_u, de_uk) values go through a 96-float per-wave LDS buffer to be broadcast to the channel
//     lanes; degrees above 8 take further passes (any degree is correct);
//   * same arithmetic, same partial-row layout and the same fixed-order two-stage reduction as the generic kernel
//     (deterministic).

__global__ __launch_bounds__(kThreads) void gatv2_bwd_pair_kernel(
    const float* __restrict__ x_src, const float* __restrict__ x_dst, const int32_t* __restrict__ seg_off, int N,
    const float* __restrict__ W_s, const float* __restrict__ b_s, const float* __restrict__ W_d,
    const float* __restrict__ b_d, const float* __restrict__ attn, float slope, const float* __restrict__ out,
    const float* __restrict__ d_out, int ld_out, const float* __restrict__ a_save, float* __restrict__ partial) {
  constexpr int FS = 2, NH = 4, D = 64, H = NH * D;
  constexpr int P = partial_len<FS>(H);
  __shared__ float sRed[P];
  __shared__ __attribute__((aligned(16))) float sX[kWavesPerBlock][2][8][2];     // x_u of (half, slot)
  __shared__ __attribute__((aligned(16))) float sDE[kWavesPerBlock][2][8][NH];   // de_uk of (half, slot, head)

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 4, half = (lane >> 3) & 1, slot = lane & 7;
  const float c_lin = 0.5f * (1.f + slope), c_abs = 0.5f * (1.f - slope);

  // per-lane constants for channels n = 4*lane + r
  float Ws[4][2], att[4], wd[4][2], bc[4];
  {
    const float4 w_lo = reinterpret_cast<const float4*>(W_s)[2 * lane], w_hi = reinterpret_cast<const float4*>(W_s)[2 * lane + 1];
    const float4 d_lo = reinterpret_cast<const float4*>(W_d)[2 * lane], d_hi = reinterpret_cast<const float4*>(W_d)[2 * lane + 1];
    const float4 a4 = reinterpret_cast<const float4*>(attn)[lane];
    const float4 bs4 = reinterpret_cast<const float4*>(b_s)[lane], bd4 = reinterpret_cast<const float4*>(b_d)[lane];
    Ws[0][0] = w_lo.x; Ws[0][1] = w_lo.y; Ws[1][0] = w_lo.z; Ws[1][1] = w_lo.w;
    Ws[2][0] = w_hi.x; Ws[2][1] = w_hi.y; Ws[3][0] = w_hi.z; Ws[3][1] = w_hi.w;
    wd[0][0] = d_lo.x; wd[0][1] = d_lo.y; wd[1][0] = d_lo.z; wd[1][1] = d_lo.w;
    wd[2][0] = d_hi.x; wd[2][1] = d_hi.y; wd[3][0] = d_hi.z; wd[3][1] = d_hi.w;
    att[0] = a4.x; att[1] = a4.y; att[2] = a4.z; att[3] = a4.w;
    bc[0] = bs4.x + bd4.x; bc[1] = bs4.y + bd4.y; bc[2] = bs4.z + bd4.z; bc[3] = bs4.w + bd4.w;
  }
  float aWs[4][2], abs_[4], aWd[4][2], abd[4], aatt[4], aWr[4][2], abr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    aWs[r][0] = aWs[r][1] = aWd[r][0] = aWd[r][1] = aWr[r][0] = aWr[r][1] = 0.f;
    abs_[r] = abd[r] = aatt[r] = abr[r] = 0.f;
  }
  float* __restrict__ xw = &sX[wave][0][0][0];
  float* __restrict__ dw = &sDE[wave][0][0][0];

  const int stride = gridDim.x * kWavesPerBlock;
  const int it0 = blockIdx.x * kWavesPerBlock + wave;
  const int NP = (N + 1) >> 1;

  // per-destination tail (lane <-> channel): fold S1 / S2 / P / Sb of ONE destination into the register accumulators
  auto finish = [&](const float (&g)[4], const float (&c)[4], const float (&S1)[4], const float (&S2)[4][2], const float P0,
                    const float P1, const float Sb0, const float Sb1, const float xv0, const float xv1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float wS2 = fmaf(Ws[r][1], S2[r][1], Ws[r][0] * S2[r][0]);
      const float wP = fmaf(Ws[r][1], P1, Ws[r][0] * P0);
      aWs[r][0] += att[r] * fmaf(c_abs, S2[r][0], c_lin * P0) + g[r] * Sb0;
      aWs[r][1] += att[r] * fmaf(c_abs, S2[r][1], c_lin * P1) + g[r] * Sb1;
      aatt[r] += fmaf(c_abs, fmaf(c[r], S1[r], wS2), c_lin * wP);
      const float der = att[r] * c_abs * S1[r];
      abs_[r] += g[r] + der;
      abd[r] += der;
      aWd[r][0] = fmaf(der, xv0, aWd[r][0]);
      aWd[r][1] = fmaf(der, xv1, aWd[r][1]);
    }
  };

  for (int kb = 0; it0 + kb * stride < NP; kb += kWave) {
    const int my_p = it0 + (kb + lane) * stride;
    const bool mine = my_p < NP;
    const int vA0 = 2 * my_p;
    const bool hasB0 = mine && (vA0 + 1 < N);
    const int m_n0 = mine ? seg_off[vA0] : 0;
    const int m_n1 = mine ? seg_off[vA0 + 1] : 0;
    const int m_n2 = hasB0 ? seg_off[vA0 + 2] : m_n1;
    const float2 m_xa = mine ? *reinterpret_cast<const float2*>(x_dst + 2 * vA0) : make_float2(0.f, 0.f);
    const float2 m_xb = hasB0 ? *reinterpret_cast<const float2*>(x_dst + 2 * vA0 + 2) : make_float2(0.f, 0.f);
    const int cnt = min(kWave, (NP - it0 - kb * stride + stride - 1) / stride);
    // Software pipeline over the pairs of this chunk: rows and first-pass edge data of pair ii+1 are requested (clamped,
    // never predicated loads: the compiler can count them) before pair ii computes - with two wavefronts per SIMD an
    // exposed HBM round trip per pair would dominate everything else.
    float4 q_oA, q_dA, q_oB, q_dB;
    float2 q_xe;
    float q_ae;
    auto request = [&](const int ix) {
      const int pp = it0 + (kb + ix) * stride;
      const int va = 2 * pp;
      const bool hb = va + 1 < N;
      const int a0 = __builtin_amdgcn_readlane(m_n0, ix), a1 = __builtin_amdgcn_readlane(m_n1, ix);
      const int a2 = __builtin_amdgcn_readlane(m_n2, ix);
      const size_t ra = static_cast<size_t>(va) * ld_out, rb = static_cast<size_t>(hb ? va + 1 : va) * ld_out;
      q_oA = *reinterpret_cast<const float4*>(out + ra + 4 * lane);
      q_dA = *reinterpret_cast<const float4*>(d_out + ra + 4 * lane);
      q_oB = *reinterpret_cast<const float4*>(out + rb + 4 * lane);
      q_dB = *reinterpret_cast<const float4*>(d_out + rb + 4 * lane);
      const int e0 = half ? a1 : a0, dg = half ? a2 - a1 : a1 - a0;
      const size_t u = static_cast<size_t>(slot < dg ? e0 + slot : 0);
      q_xe = *reinterpret_cast<const float2*>(x_src + u * FS);
      q_ae = a_save[u * NH + k];
    };
    request(0);
    for (int ii = 0; ii < cnt; ++ii) {
      const int p = it0 + (kb + ii) * stride;
      const int vA = 2 * p;
      const bool hasB = vA + 1 < N;
      const int n0 = __builtin_amdgcn_readlane(m_n0, ii), n1 = __builtin_amdgcn_readlane(m_n1, ii);
      const int n2 = __builtin_amdgcn_readlane(m_n2, ii);
      const float xa0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xa.x), ii));
      const float xa1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xa.y), ii));
      const float xb0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xb.x), ii));
      const float xb1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xb.y), ii));
      const int degA = n1 - n0, degB = n2 - n1;
      const float4 oA = q_oA, dA = q_dA, oB = q_oB, dB = q_dB;
      const int my_e0 = half ? n1 : n0, my_deg = half ? degB : degA;
      const int dmax = max(degA, degB);
      float2 xe = slot < my_deg ? q_xe : make_float2(0.f, 0.f);
      float ae = slot < my_deg ? q_ae : 0.f;
      request(min(ii + 1, cnt - 1));           // the last iteration re-requests itself (harmless, keeps the count static)
      float gA[4], gB[4], cA[4], cB[4];
      {
        const float oa[4] = {oA.x, oA.y, oA.z, oA.w}, da[4] = {dA.x, dA.y, dA.z, dA.w};
        const float ob[4] = {oB.x, oB.y, oB.z, oB.w}, db[4] = {dB.x, dB.y, dB.z, dB.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gA[r] = oa[r] > 0.f ? da[r] : 0.f;                       // ReLU mask
          gB[r] = (hasB && ob[r] > 0.f) ? db[r] : 0.f;
          aWr[r][0] = fmaf(gA[r], xa0, fmaf(gB[r], xb0, aWr[r][0]));
          aWr[r][1] = fmaf(gA[r], xa1, fmaf(gB[r], xb1, aWr[r][1]));
          abr[r] += gA[r] + gB[r];
          cA[r] = fmaf(wd[r][1], xa1, fmaf(wd[r][0], xa0, bc[r]));
          cB[r] = fmaf(wd[r][1], xb1, fmaf(wd[r][0], xb0, bc[r]));
        }
      }
      if (dmax == 0) continue;
      // G[k][f] = sum_d g[k,d] W_s[k,d,f] of both destinations: all-reduce over the 16 lanes of row k
      float GA0 = 0.f, GA1 = 0.f, GB0 = 0.f, GB1 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        GA0 = fmaf(gA[r], Ws[r][0], GA0); GA1 = fmaf(gA[r], Ws[r][1], GA1);
        GB0 = fmaf(gB[r], Ws[r][0], GB0); GB1 = fmaf(gB[r], Ws[r][1], GB1);
      }
      GA0 = row16_allsum(GA0); GA1 = row16_allsum(GA1); GB0 = row16_allsum(GB0); GB1 = row16_allsum(GB1);
      const float G0 = half ? GB0 : GA0, G1 = half ? GB1 : GA1;
      // T[k] = sum_u a_uk (G[k].x_u) over ALL in-edges of the lane's destination
      float tsum = ae * fmaf(G1, xe.y, G0 * xe.x);
      for (int base = 8; base < dmax; base += 8) {
        if (base + slot < my_deg) {
          const size_t u = static_cast<size_t>(my_e0 + base + slot);
          const float2 x2 = *reinterpret_cast<const float2*>(x_src + u * FS);
          tsum = fmaf(a_save[u * NH + k], fmaf(G1, x2.y, G0 * x2.x), tsum);
        }
      }
      const float T = half8_allsum(tsum);

      float S1A[4], S2A[4][2], S1B[4], S2B[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) S1A[r] = S2A[r][0] = S2A[r][1] = S1B[r] = S2B[r][0] = S2B[r][1] = 0.f;
      float accP0 = 0.f, accP1 = 0.f, accS0 = 0.f, accS1 = 0.f;
      for (int base = 0; base < dmax; base += 8) {
        if (base > 0) {
          const bool valid = base + slot < my_deg;
          const size_t u = static_cast<size_t>(valid ? my_e0 + base + slot : 0);
          xe = *reinterpret_cast<const float2*>(x_src + u * FS);
          ae = a_save[u * NH + k];
          if (!valid) {
            xe = make_float2(0.f, 0.f);
            ae = 0.f;
          }
        }
        const float de = ae * (fmaf(G1, xe.y, G0 * xe.x) - T);      // 0 on empty slots (a = 0)
        accP0 = fmaf(de, xe.x, accP0); accP1 = fmaf(de, xe.y, accP1);
        accS0 = fmaf(ae, xe.x, accS0); accS1 = fmaf(ae, xe.y, accS1);
        dw[(half * 8 + slot) * NH + k] = de;
        if (k == 0) *reinterpret_cast<float2*>(xw + (half * 8 + slot) * 2) = xe;
        wave_sync_lds();
        // lane <-> channel: the sign part, destination A then B (edge data = LDS broadcast within the row)
        const int cntA = min(8, degA - base), cntB = min(8, degB - base);
        for (int i = 0; i < cntA; ++i) {
          const float2 x2 = *reinterpret_cast<const float2*>(xw + i * 2);
          const float dek = dw[i * NH + k];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = fmaf(Ws[r][1], x2.y, fmaf(Ws[r][0], x2.x, cA[r]));
            const float sde = z > 0.f ? dek : -dek;
            S1A[r] += sde;
            S2A[r][0] = fmaf(sde, x2.x, S2A[r][0]);
            S2A[r][1] = fmaf(sde, x2.y, S2A[r][1]);
          }
        }
        for (int i = 0; i < cntB; ++i) {
          const float2 x2 = *reinterpret_cast<const float2*>(xw + (8 + i) * 2);
          const float dek = dw[(8 + i) * NH + k];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = fmaf(Ws[r][1], x2.y, fmaf(Ws[r][0], x2.x, cB[r]));
            const float sde = z > 0.f ? dek : -dek;
            S1B[r] += sde;
            S2B[r][0] = fmaf(sde, x2.x, S2B[r][0]);
            S2B[r][1] = fmaf(sde, x2.y, S2B[r][1]);
          }
        }
        wave_sync_lds();
      }
      // P[k][f] = sum_u de_uk x_u, Sb[k][f] = sum_u a_uk x_u per destination: all-reduce over its 8 slots, then the other
      // half's values by a row rotation
      const float p0 = half8_allsum(accP0), p1 = half8_allsum(accP1), b0 = half8_allsum(accS0), b1 = half8_allsum(accS1);
      const float q0 = dpp_f<kRowRorCtl + 8>(p0), q1 = dpp_f<kRowRorCtl + 8>(p1);
      const float r0 = dpp_f<kRowRorCtl + 8>(b0), r1 = dpp_f<kRowRorCtl + 8>(b1);
      if (degA > 0) finish(gA, cA, S1A, S2A, half ? q0 : p0, half ? q1 : p1, half ? r0 : b0, half ? r1 : b1, xa0, xa1);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { /* isolated destination: only the residual gradient (done above) */ }
      }
      if (degB > 0) finish(gB, cB, S1B, S2B, half ? p0 : q0, half ? p1 : q1, half ? b0 : r0, half ? b1 : r1, xb0, xb1);
    }
  }

  // fold the 4 waves in fixed order through LDS, then one partial row per workgroup (layout of the generic kernel)
  for (int w = 0; w < kWavesPerBlock; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 4 * lane + r;
        auto put = [&](int idx, float val) { sRed[idx] = (w == 0) ? val : sRed[idx] + val; };
        put(n * FS + 0, aWs[r][0]);
        put(n * FS + 1, aWs[r][1]);
        int o = H * FS;
        put(o + n, abs_[r]);
        o += H;
        put(o + 2 * n, aWd[r][0]);
        put(o + 2 * n + 1, aWd[r][1]);
        o += 2 * H;
        put(o + n, abd[r]);
        o += H;
        put(o + n, aatt[r]);
        o += H;
        put(o + 2 * n, aWr[r][0]);
        put(o + 2 * n + 1, aWr[r][1]);
        o += 2 * H;
        put(o + n, abr[r]);
      }
    }
    __syncthreads();
  }
  float* __restrict__ prow = partial + static_cast<size_t>(blockIdx.x) * P;
  for (int i = tid; i < P; i += kThreads) prow[i] = sRed[i];
}


template <int FS, int NH, int D>
int launch_fwd(const float* x_src, const float* x_dst, const int32_t* seg_off, const int32_t* dst_order, int N,
               const float* W_s,
               const float* b_s, const float* W_d, const float* b_d, const float* attn, const float* W_r,
               const float* b_r, float slope, float* out, int ld_out, float* a_save, hipStream_t st) {
  const int grid = capped_grid(N, kWavesPerBlock, 4096);
  hipLaunchKernelGGL((gatv2_fwd_kernel<FS, NH, D>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off, dst_order,
                     N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope, out, ld_out, a_save);
  return launch_status();
}

template <int FS, int NH, int D>
int launch_bwd(const float* x_src, const float* x_dst, const int32_t* seg_off, const int32_t* dst_order, int N,
               const float* W_s,
               const float* b_s, const float* W_d, const float* b_d, const float* attn, float slope, const float* out,
               const float* d_out, int ld_out, const float* a_save, const GradPtrs& gp, float* ws, bool sparse_hint,
               bool mfma, hipStream_t st) {
  constexpr int H = NH * D;
  constexpr int P = partial_len<FS>(H);
  const int grid = bwd_blocks(N);
  // A/B switch: UAVGNN_BWD_ONEPASS=0 sends every destination through the general two-pass path (same bits out)
  static const int onepass = (getenv("UAVGNN_BWD_ONEPASS") && getenv("UAVGNN_BWD_ONEPASS")[0] == '0') ? 0 : 2 * kWave;
  if constexpr (FS == 4 && NH == 4 && D == 64) {
    if (mfma && onepass > 0) {
      const int g = capped_grid(N, kWavesPerBlock, kMaxBwdBlocks / 2);
      int rc = gatv2_bwd_mfma_launch(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, slope, out, d_out, ld_out,
                                     a_save, ws, onepass, g, st);
      if (rc) return rc;
      hipLaunchKernelGGL((gatv2_bwd_kernel<FS, NH, D, true, false>), dim3(g), dim3(kThreads), 0, st, x_src, x_dst, seg_off, dst_order,
                         N, W_s, b_s, W_d, b_d, attn, slope, out, d_out, ld_out, a_save, ws + static_cast<size_t>(g) * P, onepass, 2);
      rc = launch_status();
      if (rc) return rc;
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((P + kWave - 1) / kWave), dim3(1024), 0, st, ws, 2 * g, P, gp);
      return launch_status();
    }
  }
  if (sparse_hint)
    hipLaunchKernelGGL((gatv2_bwd_kernel<FS, NH, D, true>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off,
                       dst_order, N, W_s, b_s, W_d, b_d, attn, slope, out, d_out, ld_out, a_save, ws, onepass, 0);
  else
    hipLaunchKernelGGL((gatv2_bwd_kernel<FS, NH, D, false>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off,
                       dst_order, N, W_s, b_s, W_d, b_d, attn, slope, out, d_out, ld_out, a_save, ws, onepass, 0);
  int rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((P + kWave - 1) / kWave), dim3(1024), 0, st, ws, grid, P, gp);
  return launch_status();
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

#define UAVGNN_DISPATCH(FSV, NHV, DV, CALL)                         \
  if (F_src == FSV && nh == NHV && D == DV) {                       \
    constexpr int FS_ = FSV, NH_ = NHV, D_ = DV;                    \
    (void)FS_; (void)NH_; (void)D_;                                 \
    return CALL;                                                    \
  }

#define UAVGNN_DISPATCH_ALL(CALL)    \
  UAVGNN_DISPATCH(4, 4, 64, CALL)    \
  UAVGNN_DISPATCH(2, 4, 64, CALL)    \
  UAVGNN_DISPATCH(4, 4, 16, CALL)    \
  UAVGNN_DISPATCH(2, 4, 16, CALL)    \
  UAVGNN_DISPATCH(4, 4, 8, CALL)     \
  UAVGNN_DISPATCH(2, 4, 8, CALL)     \
  UAVGNN_DISPATCH(4, 4, 32, CALL)    \
  UAVGNN_DISPATCH(2, 4, 32, CALL)    \
  UAVGNN_DISPATCH(4, 8, 32, CALL)    \
  UAVGNN_DISPATCH(2, 8, 32, CALL)    \
  UAVGNN_DISPATCH(4, 2, 64, CALL)    \
  UAVGNN_DISPATCH(2, 2, 64, CALL)    \
  UAVGNN_DISPATCH(4, 1, 64, CALL)    \
  UAVGNN_DISPATCH(2, 1, 64, CALL)

// variant: 0 = VALU kernel only, 1 = MFMA (row-tile) kernel when instantiated, 2 = automatic (low-degree kernel for
// two-feature relations with mean in-degree <= 8, else MFMA, else VALU)
static int gatv2_fwd_checked(int variant, const float* x_src, int E, int F_src, const float* x_dst, int F_dst,
                             const int32_t* seg_off, const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d,
                             const float* b_d, const float* attn, const float* W_r, const float* b_r, int nh, int D,
                             float slope, float* out, int ld_out, float* attn_save, uavgnn_stream_t stream) {
  if (N < 0 || E < 0 || (E > 0 && !x_src) || !seg_off || !x_dst || !W_s || !b_s || !W_d || !b_d || !attn || !W_r ||
      !out || ld_out < nh * D)
    return UAVGNN_EINVAL;
  if (F_dst != 2) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (variant == 2) {
    const int rc = gatv2_fwd_small(F_src, nh, D, x_src, E, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r,
                                   slope, out, ld_out, attn_save, st);
    if (rc != UAVGNN_EUNSUPPORTED) return rc;
  }
  if (variant >= 1) {
    const int rc = gatv2_fwd_mfma(F_src, nh, D, x_src, E, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope,
                                  out, ld_out, attn_save, st);
    if (rc != UAVGNN_EUNSUPPORTED) return rc;
  }
  UAVGNN_DISPATCH_ALL((launch_fwd<FS_, NH_, D_>(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope,
                                                 out, ld_out, attn_save, st)))
  return UAVGNN_EUNSUPPORTED;
}

extern "C" int uavgnn_gatv2_fwd(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                                const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d,
                                const float* attn, const float* W_r, const float* b_r, int nh, int D, float slope,
                                float* out, int ld_out, float* attn_save, uavgnn_stream_t stream) {
  return gatv2_fwd_checked(2, x_src, E, F_src, x_dst, F_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, nh, D,
                           slope, out, ld_out, attn_save, stream);
}

extern "C" int uavgnn_gatv2_fwd_mfma(const float* x_src, int E, int F_src, const float* x_dst, int F_dst,
                                     const int32_t* seg_off, const int32_t* dst_order, int N, const float* W_s, const float* b_s,
                                     const float* W_d, const float* b_d, const float* attn, const float* W_r,
                                     const float* b_r, int nh, int D, float slope, float* out, int ld_out,
                                     float* attn_save, uavgnn_stream_t stream) {
  return gatv2_fwd_checked(1, x_src, E, F_src, x_dst, F_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, nh, D,
                           slope, out, ld_out, attn_save, stream);
}

extern "C" int uavgnn_gatv2_fwd_valu(const float* x_src, int E, int F_src, const float* x_dst, int F_dst,
                                     const int32_t* seg_off, const int32_t* dst_order, int N, const float* W_s, const float* b_s,
                                     const float* W_d, const float* b_d, const float* attn, const float* W_r,
                                     const float* b_r, int nh, int D, float slope, float* out, int ld_out,
                                     float* attn_save, uavgnn_stream_t stream) {
  return gatv2_fwd_checked(0, x_src, E, F_src, x_dst, F_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, nh, D,
                           slope, out, ld_out, attn_save, stream);
}

extern "C" size_t uavgnn_gatv2_bwd_workspace_bytes(int F_src, int H) {
  return static_cast<size_t>(kMaxBwdBlocks) * static_cast<size_t>(H) * (F_src + 8) * sizeof(float);
}

// variant: 0 = generic kernel only (one destination per wavefront), 1 = automatic (pair kernel for low-degree two-feature
// relations, else generic)
static int gatv2_bwd_checked(int variant, const float* x_src, int E, int F_src, const float* x_dst, int F_dst,
                             const int32_t* seg_off, const int32_t* dst_order, int N, const float* W_s, const float* b_s,
                             const float* W_d, const float* b_d, const float* attn, int nh, int D, float slope,
                             const float* out, const float* d_out, int ld_out, const float* attn_save, float* dW_s,
                             float* db_s, float* dW_d, float* db_d, float* dattn, float* dW_r, float* db_r,
                             void* workspace, size_t workspace_bytes, uavgnn_stream_t stream) {
  if (N <= 0 || !seg_off || !x_dst || !W_s || !b_s || !W_d || !b_d || !attn || !out || !d_out || !attn_save ||
      !dW_s || !db_s || !dW_d || !db_d || !dattn || !dW_r || !db_r || !workspace)
    return UAVGNN_EINVAL;
  if (F_dst != 2) return UAVGNN_EUNSUPPORTED;
  const int H = nh * D;
  if (workspace_bytes < uavgnn_gatv2_bwd_workspace_bytes(F_src, H)) return UAVGNN_EWORKSPACE;
  GradPtrs gp;
  gp.p[0] = dW_s; gp.p[1] = db_s; gp.p[2] = dW_d; gp.p[3] = db_d; gp.p[4] = dattn; gp.p[5] = dW_r; gp.p[6] = db_r;
  gp.off[0] = 0;
  gp.off[1] = H * F_src;
  gp.off[2] = gp.off[1] + H;
  gp.off[3] = gp.off[2] + 2 * H;
  gp.off[4] = gp.off[3] + H;
  gp.off[5] = gp.off[4] + H;
  gp.off[6] = gp.off[5] + 2 * H;
  gp.off[7] = gp.off[6] + H;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = static_cast<float*>(workspace);
  const bool sparse_hint = static_cast<long long>(E) < 16LL * N;
  // low-degree two-feature relations (`near`): two destinations per wavefront (any degree is correct; pays for mean <= 8)
  if (variant == 1 && F_src == 2 && nh == 4 && D == 64 && E > 0 && static_cast<long long>(E) <= 8LL * N && (ld_out % 4) == 0 &&
      ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(W_s) |
        reinterpret_cast<uintptr_t>(W_d) | reinterpret_cast<uintptr_t>(b_s) | reinterpret_cast<uintptr_t>(b_d) |
        reinterpret_cast<uintptr_t>(attn)) & 15) == 0 && (reinterpret_cast<uintptr_t>(x_src) & 7) == 0) {
    const int grid = bwd_blocks((N + 1) / 2);
    hipLaunchKernelGGL(gatv2_bwd_pair_kernel, dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off, N, W_s, b_s, W_d,
                       b_d, attn, slope, out, d_out, ld_out, attn_save, ws);
    int rc = launch_status();
    if (rc) return rc;
    constexpr int Pn = partial_len<2>(256);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((Pn + kWave - 1) / kWave), dim3(1024), 0, st, ws, grid, Pn, gp);
    return launch_status();
  }
  // dense batches (mean in-degree >= 16) of the flagship shape: destinations with 16 .. 128 in-edges on the matrix cores, the
  // rest in a second launch of the packed-FMA kernel.  A/B switch: UAVGNN_K1_BWD_MFMA=0 (same results up to sign ties of z).
  static const bool mfma_on = !(getenv("UAVGNN_K1_BWD_MFMA") && getenv("UAVGNN_K1_BWD_MFMA")[0] == '0');
  const bool use_mfma = variant == 1 && mfma_on && !sparse_hint && (reinterpret_cast<uintptr_t>(x_src) & 15) == 0;
  UAVGNN_DISPATCH_ALL((launch_bwd<FS_, NH_, D_>(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, slope, out, d_out,
                                                 ld_out, attn_save, gp, ws, sparse_hint,
                                                 use_mfma, st)))
  return UAVGNN_EUNSUPPORTED;
}

extern "C" int uavgnn_gatv2_bwd(const float* x_src, int E, int F_src, const float* x_dst, int F_dst, const int32_t* seg_off,
                                const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d, const float* b_d,
                                const float* attn, int nh, int D, float slope, const float* out, const float* d_out,
                                int ld_out, const float* attn_save, float* dW_s, float* db_s, float* dW_d, float* db_d,
                                float* dattn, float* dW_r, float* db_r, void* workspace, size_t workspace_bytes,
                                uavgnn_stream_t stream) {
  return gatv2_bwd_checked(1, x_src, E, F_src, x_dst, F_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, nh, D, slope, out,
                           d_out, ld_out, attn_save, dW_s, db_s, dW_d, db_d, dattn, dW_r, db_r, workspace, workspace_bytes,
                           stream);
}

extern "C" int uavgnn_gatv2_bwd_generic(const float* x_src, int E, int F_src, const float* x_dst, int F_dst,
                                        const int32_t* seg_off, const int32_t* dst_order, int N, const float* W_s,
                                        const float* b_s, const float* W_d, const float* b_d, const float* attn, int nh,
                                        int D, float slope, const float* out, const float* d_out, int ld_out,
                                        const float* attn_save, float* dW_s, float* db_s, float* dW_d, float* db_d,
                                        float* dattn, float* dW_r, float* db_r, void* workspace, size_t workspace_bytes,
                                        uavgnn_stream_t stream) {
  return gatv2_bwd_checked(0, x_src, E, F_src, x_dst, F_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, nh, D, slope, out,
                           d_out, ld_out, attn_save, dW_s, db_s, dW_d, db_d, dattn, dW_r, db_r, workspace, workspace_bytes,
                           stream);
}
#endif  // UAVGNN_GATV2_BWD_MFMA_TU
