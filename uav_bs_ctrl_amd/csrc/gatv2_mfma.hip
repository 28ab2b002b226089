// K1 forward, fp32-MFMA variant (nh = 4, D in {16,32,64}): the hot kernel of the path.
//
// Same contract as gatv2_fwd_kernel in gatv2.hip (reference: dglnn.GATv2Conv.forward as used at
// /root/reference/algos/madrqn/agents/gnn_agents.py:93-96,:103-104; math SURVEY Appendix A.1/A.3).
//
// One wavefront owns one destination at a time.  Its in-edges are processed 16 at a time ("row tile"):
//   Z^T[16 channels x 16 edges] = W_s[16 x F] . X^T[F x 16]  + c[v, channels]      v_mfma_f32_16x16x4_f32, K = F exactly;
//       A operand = one weight per lane (register-resident for all H/16 channel tiles),
//       B operand = the edge's input feature, ONE dword per lane straight from global memory,
//       C operand = the destination term b_s + W_d x_v + b_d, register-resident per destination;
//   lrelu(z) = (1+s)/2 z + (1-s)/2 |z|  (Appendix A.3 i): the linear half is a 4-float dot per (edge, head), the |z| half is
//       one |.|-source-modifier FMA per channel on the MFMA result - the only per-(edge,channel) VALU work;
//   the per-head score needs a sum over channels = over the 4 lane groups (lane>>4) of the MFMA D layout:
//       two v_permlane32_swap + one v_permlane16_swap + 3 adds leave  e[edge = lane&15][head = lane>>4]  - one per lane;
//   online softmax state per lane, combined across the 16 lanes of a head once per destination; the aggregate is taken in
//   input space (Appendix A.3 ii) and projected once per destination in the lane<->channel epilogue (coalesced row stores).
// fp32 MFMA is bit-for-bit an fmaf chain (guide section 3), so numerics equal the VALU kernel's up to summation order.
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int NH = 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// All-reduce over the 16 lanes of a row (lane>>4 fixed) with DPP row rotations: v += ror(v, 8), 4, 2, 1.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
constexpr int kRowRor = 0x120;  // DPP control base of row_ror:n
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

// Sum pe[k] over the four 16-lane groups and leave head (lane>>4)'s total in every lane: 3 swaps + 3 adds.
__device__ __forceinline__ float reduce_heads(float pe0, float pe1, float pe2, float pe3) {
  auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe0), __float_as_uint(pe2), false, false);
  const float a = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);  // lo half: head 0 over {g,g+2}; hi half: head 2
  auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe1), __float_as_uint(pe3), false, false);
  const float b = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);  // lo half: head 1; hi half: head 3
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

template <int FS>
__device__ __forceinline__ void load_xrow(const float* __restrict__ p, bool valid, float (&x)[FS]) {
  if (valid) {
    if constexpr (FS == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
      const float2 t = *reinterpret_cast<const float2*>(p);
      x[0] = t.x; x[1] = t.y;
    }
  } else {
#pragma unroll
    for (int f = 0; f < FS; ++f) x[f] = 0.f;
  }
}

constexpr float kLog2e = 1.4426950408889634f;

// CHUNK selects how destinations are handed to the per-destination code:
//   false - scalar loads of (id, offsets, x_v) one iteration ahead: best when every destination carries real work;
//   true  - 64 destinations' meta data by VECTOR loads (lane <-> destination) + v_readlane: no scalar-load latency
//           chain, which bounds batches where most agents see nothing (94 % under a random policy): 62 % vs 44 % of
//           the HBM peak on an all-zero-degree batch of 262 144 agents.
template <int FS, int D, bool CHUNK>
__global__ __launch_bounds__(kThreads, 2) void gatv2_fwd_mfma_kernel(
    const float* __restrict__ x_src, const float* __restrict__ x_dst, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ dst_order, int N, const float* __restrict__ W_s, const float* __restrict__ b_s,
    const float* __restrict__ W_d, const float* __restrict__ b_d, const float* __restrict__ attn,
    const float* __restrict__ W_r, const float* __restrict__ b_r, float slope, float* __restrict__ out, int ld_out,
    float* __restrict__ a_save) {
  constexpr int H = NH * D;
  constexpr int CT = H / 16;        // channel tiles
  constexpr int TPH = D / 16;       // channel tiles per head
  constexpr int J = (H + kWave - 1) / kWave;
  __shared__ float sW[H * FS];
  __shared__ float sAttn[H];
  __shared__ float sWa[NH * 4];     // wa[k][f] = sum_d attn[k,d] W_s[k,d,f]
  __shared__ __attribute__((aligned(16))) float sC[kWavesPerBlock][H];
  __shared__ float sS[kWavesPerBlock][NH * FS];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15;          // edge within the row tile (MFMA column)
  const int g = lane >> 4;          // lane group: input feature for A/B operands, head after the reduction

  for (int i = tid; i < H * FS; i += kThreads) sW[i] = W_s[i];
  for (int i = tid; i < H; i += kThreads) sAttn[i] = attn[i];
  __syncthreads();
  {  // wa[k][f]: 16 outputs x 16 partial sums each, folded with row rotations (256 threads = 16 rows of 16 lanes)
    const int kf = tid >> 4, part = tid & 15;
    const int k = kf >> 2, f = kf & 3;
    float acc = 0.f;
    if (f < FS)
      for (int d = part; d < D; d += 16) acc = fmaf(sAttn[k * D + d], sW[(k * D + d) * FS + f], acc);
    acc = row16_sum(acc);
    if (part == 0) sWa[kf] = acc;
  }

  // ---- per-lane constants ------------------------------------------------------------------------------------
  // Scores are kept in the log2 domain (log2(e) folded into the attention vector) so the softmax uses v_exp_f32.
  float Wa[CT];        // A operand of tile ct: W_s[ct*16 + j][g]
  float att[CT][4];    // log2e * (1-s)/2 * attn[ct*16 + 4g + r]   (D-layout rows of this lane)
  const float c_abs = kLog2e * 0.5f * (1.f - slope), c_lin = kLog2e * 0.5f * (1.f + slope);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    Wa[ct] = (g < FS) ? sW[(ct * 16 + j) * FS + g] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) att[ct][r] = c_abs * sAttn[ct * 16 + 4 * g + r];
  }
  float wd0[J], wd1[J], bc[J], wr0[J], wr1[J], br[J], bs[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int n = lane + kWave * jj;
    const bool ok = n < H;
    wd0[jj] = ok ? W_d[n * 2 + 0] : 0.f;
    wd1[jj] = ok ? W_d[n * 2 + 1] : 0.f;
    bs[jj] = ok ? b_s[n] : 0.f;
    bc[jj] = ok ? b_d[n] + bs[jj] : 0.f;
    wr0[jj] = ok ? W_r[n * 2 + 0] : 0.f;
    wr1[jj] = ok ? W_r[n * 2 + 1] : 0.f;
    br[jj] = (ok && b_r != nullptr) ? b_r[n] : 0.f;
  }
  __syncthreads();
  float wlin[NH];      // log2e * (1+s)/2 * wa[k][g]: this lane's share of the linear half of the score
#pragma unroll
  for (int k = 0; k < NH; ++k) wlin[k] = c_lin * sWa[k * 4 + g];

  float* __restrict__ cw = sC[wave];
  float* __restrict__ sw = sS[wave];

  const int stride = gridDim.x * kWavesPerBlock;
  const int it0 = blockIdx.x * kWavesPerBlock + wave;   // this wave owns positions it0 + k*stride of the hand-out order

  auto process = [&](const int v, const int ce0, const int cdeg, const float cxv0, const float cxv1) {
    // first row tile of this destination: issue the loads before anything else
    float xr[FS];
    load_xrow<FS>(x_src + static_cast<size_t>(ce0 + j) * FS, j < cdeg, xr);
    float xBn = (j < cdeg && g < FS) ? x_src[static_cast<size_t>(ce0 + j) * FS + g] : 0.f;
    float res[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) res[jj] = fmaf(wr1[jj], cxv1, fmaf(wr0[jj], cxv0, br[jj]));
    float* __restrict__ orow = out + static_cast<size_t>(v) * ld_out;
    if (cdeg == 0) {
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        const int n = lane + kWave * jj;
        if (n < H) orow[n] = fmaxf(res[jj], 0.f);
      }
      return;
    }
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int n = lane + kWave * jj;
      if (n < H) cw[n] = fmaf(wd1[jj], cxv1, fmaf(wd0[jj], cxv0, bc[jj]));
    }
    wave_sync_lds();
    f32x4 cinit[CT];   // C operand: destination term for channels ct*16 + 4g + r
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) cinit[ct] = *reinterpret_cast<const f32x4*>(cw + ct * 16 + 4 * g);

    float m = -INFINITY, den = 0.f, s[FS];
#pragma unroll
    for (int f = 0; f < FS; ++f) s[f] = 0.f;

    for (int base = 0; base < cdeg; base += 16) {
      const bool valid = base + j < cdeg;
      const size_t u = static_cast<size_t>(ce0 + base + j);
      float xc[FS];
#pragma unroll
      for (int f = 0; f < FS; ++f) xc[f] = xr[f];
      const float xB = xBn;
      // next row tile's inputs are in flight while this tile computes
      const bool nvalid = base + 16 + j < cdeg;
      load_xrow<FS>(x_src + (u + 16) * FS, nvalid, xr);
      xBn = (nvalid && g < FS) ? x_src[(u + 16) * FS + g] : 0.f;
      float pe[NH][2];
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        pe[k][0] = wlin[k] * xB;
        pe[k][1] = 0.f;
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const f32x4 z = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[ct], xB, cinit[ct], 0, 0, 0);
        const int k = ct / TPH;
        pe[k][0] = fmaf(att[ct][0], fabsf(z[0]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][1], fabsf(z[1]), pe[k][1]);
        pe[k][0] = fmaf(att[ct][2], fabsf(z[2]), pe[k][0]);
        pe[k][1] = fmaf(att[ct][3], fabsf(z[3]), pe[k][1]);
      }
      // log2-domain score of (edge j, head g), up to a constant per (v, head) that cancels in the softmax
      const float e = reduce_heads(pe[0][0] + pe[0][1], pe[1][0] + pe[1][1], pe[2][0] + pe[2][1], pe[3][0] + pe[3][1]);
      if (valid) {
        if (a_save != nullptr) a_save[u * NH + g] = e;
        const float mn = fmaxf(m, e);
        const float sc = __builtin_amdgcn_exp2f(m - mn);   // exp2(-inf) = 0 on the first edge
        const float p = __builtin_amdgcn_exp2f(e - mn);
        den = fmaf(den, sc, p);
#pragma unroll
        for (int f = 0; f < FS; ++f) s[f] = fmaf(s[f], sc, p * xc[f]);
        m = mn;
      }
    }
    // ---- combine the 16 lanes of each head ------------------------------------------------------------------
    const float mx = row16_max(m);
    const float scl = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mx);
    const float inv = __builtin_amdgcn_rcpf(row16_sum(den * scl));
#pragma unroll
    for (int f = 0; f < FS; ++f) {
      const float t = row16_sum(s[f] * scl);
      if (j == 0) sw[g * FS + f] = t * inv;
    }
    if (a_save != nullptr) {
      for (int base = 0; base < cdeg; base += 16) {
        if (base + j < cdeg) {
          float* ap = a_save + static_cast<size_t>(ce0 + base + j) * NH + g;
          *ap = __builtin_amdgcn_exp2f(*ap - mx) * inv;
        }
      }
    }
    wave_sync_lds();
    // ---- epilogue: lane <-> channel --------------------------------------------------------------------------
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int n = lane + kWave * jj;
      if (n < H) {
        const int k = n / D;
        float agg = bs[jj];
#pragma unroll
        for (int f = 0; f < FS; ++f) agg = fmaf(sW[n * FS + f], sw[k * FS + f], agg);
        orow[n] = fmaxf(agg + res[jj], 0.f);
      }
    }
    wave_sync_lds();
  };

  if constexpr (CHUNK) {
    for (int kb = 0; it0 + kb * stride < N; kb += kWave) {
      const int my_it = it0 + (kb + lane) * stride;
      const bool mine = my_it < N;
      const int m_v = mine ? (dst_order ? dst_order[my_it] : my_it) : 0;
      const int m_e0 = mine ? seg_off[m_v] : 0;
      const int m_e1 = mine ? seg_off[m_v + 1] : 0;
      const float2 m_xv = mine ? *reinterpret_cast<const float2*>(x_dst + 2 * m_v) : make_float2(0.f, 0.f);
      const int cnt = min(kWave, (N - it0 - kb * stride + stride - 1) / stride);
      for (int ii = 0; ii < cnt; ++ii) {
        const int e0 = __builtin_amdgcn_readlane(m_e0, ii);
        process(__builtin_amdgcn_readlane(m_v, ii), e0, __builtin_amdgcn_readlane(m_e1, ii) - e0,
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xv.x), ii)),
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(m_xv.y), ii)));
      }
    }
  } else {
    if (it0 >= N) return;
    // destination scalars are fetched one iteration ahead (raw values: nothing derived from them until next iteration)
    int nv = dst_order ? dst_order[it0] : it0;
    float2 nxv = *reinterpret_cast<const float2*>(x_dst + 2 * nv);
    int ne0 = seg_off[nv], ne1 = seg_off[nv + 1];
    int nnv = dst_order ? dst_order[min(it0 + stride, N - 1)] : min(it0 + stride, N - 1);
    for (int it = it0; it < N; it += stride) {
      const int v = nv, ce0 = ne0, cdeg = ne1 - ne0;
      const float cxv0 = nxv.x, cxv1 = nxv.y;
      {  // clamped look-ahead: always legal addresses, unused after the last iteration
        nv = nnv;
        nxv = *reinterpret_cast<const float2*>(x_dst + 2 * nv);
        ne0 = seg_off[nv];
        ne1 = seg_off[nv + 1];
        const int it2 = min(it + 2 * stride, N - 1);
        nnv = dst_order ? dst_order[it2] : it2;
      }
      process(v, ce0, cdeg, cxv0, cxv1);
    }
  }
}

template <int FS, int D>
int launch(const float* x_src, const float* x_dst, const int32_t* seg_off, const int32_t* dst_order, int N,
           const float* W_s, const float* b_s,
           const float* W_d, const float* b_d, const float* attn, const float* W_r, const float* b_r, float slope,
           float* out, int ld_out, float* a_save, bool sparse_hint, hipStream_t st) {
  const int grid = capped_grid(N, kWavesPerBlock, 512);  // persistent: 2 workgroups per CU, constants loaded once
  if (sparse_hint)
    hipLaunchKernelGGL((gatv2_fwd_mfma_kernel<FS, D, true>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off,
                       dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope, out, ld_out, a_save);
  else
    hipLaunchKernelGGL((gatv2_fwd_mfma_kernel<FS, D, false>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off,
                       dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope, out, ld_out, a_save);
  return launch_status();
}

}  // namespace

int gatv2_fwd_mfma(int F_src, int nh, int D, const float* x_src, int E, const float* x_dst, const int32_t* seg_off,
                   const int32_t* dst_order, int N,
                   const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                   const float* W_r, const float* b_r, float slope, float* out, int ld_out, float* a_save,
                   hipStream_t st) {
  if (nh != NH) return UAVGNN_EUNSUPPORTED;
  const bool sparse_hint = static_cast<long long>(E) < 16LL * N;   // mean in-degree below one row tile
#define UAVGNN_MFMA_CASE(FSV, DV)                                                                                  \
  if (F_src == FSV && D == DV)                                                                                     \
    return launch<FSV, DV>(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope, out, ld_out, a_save,  \
                           sparse_hint, st);
  UAVGNN_MFMA_CASE(4, 64)
  UAVGNN_MFMA_CASE(2, 64)
  UAVGNN_MFMA_CASE(4, 32)
  UAVGNN_MFMA_CASE(2, 32)
  UAVGNN_MFMA_CASE(4, 16)
  UAVGNN_MFMA_CASE(2, 16)
#undef UAVGNN_MFMA_CASE
  return UAVGNN_EUNSUPPORTED;
}

}  // namespace uavgnn
