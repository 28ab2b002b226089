// K1 forward for LOW-DEGREE relations with two source features (`ubs -near-> agent`: in-degree <= n_agents - 1):
// nh = 4, D in {16, 32, 64}, F_src = 2.
//
// Same contract as gatv2_fwd_kernel / gatv2_fwd_mfma_kernel (reference: dglnn.GATv2Conv.forward as used at
// /root/reference/algos/madrqn/agents/gnn_agents.py:93-96,:103-104; math SURVEY Appendix A.1/A.3).  The MFMA kernel
// spends a whole 16-edge x 4-feature row tile (16 MFMAs) plus ~1500 cycles of per-destination bookkeeping on a
// destination with 7 edges of 2 features; here the wavefront is laid out for 8 edges at a time:
//     lane = 16 * head + 2 * edge_slot + half      (4 heads x 8 edge slots x 2 halves of the head's D channels)
// so that  * the per-(edge, channel) work is D/2 channels per lane, weights register-resident,
//          * the head's score = the two halves added with ONE quad-perm DPP,
//          * the softmax over the 8 edge slots = DPP row rotations by 8, 4, 2 (parity-preserving) inside the 16-lane row,
//          * nothing crosses rows: no permlane, no LDS in the edge loop.
// lrelu(z) = (1+s)/2 z + (1-s)/2 |z|: the linear half is a 2-float dot per (edge, head), the |z| half one
// |.|-modifier FMA per channel; scores live in the log2 domain (v_exp_f32); the aggregate is taken in input space
// (sum_e a_e x_e, 2 floats per head) and projected in a lane <-> channel epilogue with coalesced row stores.
// In-degrees above 8 are handled by further passes of 8 edges with an online softmax, so the kernel is correct for any
// degree - the dispatcher only picks it when the mean in-degree is at most 8.
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int NH = 4;
constexpr int kSlots = 8;          // edges per pass
constexpr float kLog2e = 1.4426950408889634f;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
constexpr int kRowRor = 0x120;     // DPP control base of row_ror:n
constexpr int kQuadXor1 = 0xB1;    // quad_perm:[1,0,3,2]

// all-reduce over the 8 lanes of a 16-lane row that share this lane's parity (the 8 edge slots of one (head, half))
__device__ __forceinline__ float slots_sum(float v) {
  v += dpp_mov<kRowRor + 8>(v);
  v += dpp_mov<kRowRor + 4>(v);
  v += dpp_mov<kRowRor + 2>(v);
  return v;
}
__device__ __forceinline__ float slots_max(float v) {
  v = fmaxf(v, dpp_mov<kRowRor + 8>(v));
  v = fmaxf(v, dpp_mov<kRowRor + 4>(v));
  v = fmaxf(v, dpp_mov<kRowRor + 2>(v));
  return v;
}

template <int D>
__global__ __launch_bounds__(kThreads, 2) void gatv2_fwd_small_kernel(
    const float* __restrict__ x_src, const float* __restrict__ x_dst, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ dst_order, int N, const float* __restrict__ W_s, const float* __restrict__ b_s,
    const float* __restrict__ W_d, const float* __restrict__ b_d, const float* __restrict__ attn,
    const float* __restrict__ W_r, const float* __restrict__ b_r, float slope, float* __restrict__ out, int ld_out,
    float* __restrict__ a_save) {
  constexpr int FS = 2;
  constexpr int H = NH * D;
  constexpr int CPL = D / 2;                 // channels per lane in the edge loop
  constexpr int GST = CPL + 4;               // LDS stride of a (head, half) group: conflict-free 16-byte reads
  constexpr int J = (H + kWave - 1) / kWave; // channels per lane in the epilogue
  __shared__ float sWa[NH * FS];             // wa[k][f] = sum_d attn[k,d] W_s[k,d,f]
  __shared__ __attribute__((aligned(16))) float sC[kWavesPerBlock][2 * NH * GST];
  __shared__ float sS[kWavesPerBlock][NH * FS];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 4;                   // head
  const int es = (lane >> 1) & 7;            // edge slot
  const int half = lane & 1;
  const int grp = 2 * k + half;              // (head, half) group: channels grp * CPL .. + CPL - 1

  if (tid < NH * FS) {
    const int kk = tid / FS, f = tid - kk * FS;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(attn[kk * D + d], W_s[(kk * D + d) * FS + f], acc);
    sWa[tid] = acc;
  }
  // ---- per-lane constants --------------------------------------------------------------------------------------
  const float c_abs = kLog2e * 0.5f * (1.f - slope), c_lin = kLog2e * 0.5f * (1.f + slope);
  float w0[CPL], w1[CPL], at[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = grp * CPL + i;
    w0[i] = W_s[c * FS + 0];
    w1[i] = W_s[c * FS + 1];
    at[i] = c_abs * attn[c];
  }
  float wd0[J], wd1[J], bc[J], wr0[J], wr1[J], br[J], bs[J], ws0[J], ws1[J];
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
    ws0[jj] = ok ? W_s[n * FS + 0] : 0.f;
    ws1[jj] = ok ? W_s[n * FS + 1] : 0.f;
  }
  __syncthreads();
  const float wl0 = c_lin * sWa[k * FS + 0], wl1 = c_lin * sWa[k * FS + 1];

  float* __restrict__ cw = sC[wave];
  float* __restrict__ sw = sS[wave];
  const int stride = gridDim.x * kWavesPerBlock;
  const int it0 = blockIdx.x * kWavesPerBlock + wave;

  auto process = [&](const int v, const int ce0, const int cdeg, const float cxv0, const float cxv1) {
    // first pass's edge features: issued before anything else
    float2 xe = make_float2(0.f, 0.f);
    if (es < cdeg) xe = *reinterpret_cast<const float2*>(x_src + static_cast<size_t>(ce0 + es) * FS);
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
    // destination term b_s + W_d x_v + b_d: lane <-> channel, then through LDS into the (head, half) layout
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int n = lane + kWave * jj;
      if (n < H) cw[(n / CPL) * GST + (n % CPL)] = fmaf(wd1[jj], cxv1, fmaf(wd0[jj], cxv0, bc[jj]));
    }
    wave_sync_lds();
    float dt[CPL];
#pragma unroll
    for (int i = 0; i < CPL; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(cw + grp * GST + i);
      dt[i] = t.x; dt[i + 1] = t.y; dt[i + 2] = t.z; dt[i + 3] = t.w;
    }
    float m = -INFINITY, den = 0.f, s0 = 0.f, s1 = 0.f;   // identical in the 16 lanes of a head after every pass
    for (int base = 0; base < cdeg; base += kSlots) {
      const bool valid = base + es < cdeg;
      const float x0 = xe.x, x1 = xe.y;
      const bool nvalid = base + kSlots + es < cdeg;
      if (nvalid) xe = *reinterpret_cast<const float2*>(x_src + static_cast<size_t>(ce0 + base + kSlots + es) * FS);
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; i += 2) {
        const float za = fmaf(w0[i], x0, fmaf(w1[i], x1, dt[i]));
        const float zb = fmaf(w0[i + 1], x0, fmaf(w1[i + 1], x1, dt[i + 1]));
        pa = fmaf(at[i], fabsf(za), pa);
        pb = fmaf(at[i + 1], fabsf(zb), pb);
      }
      float e = pa + pb;
      e += dpp_mov<kQuadXor1>(e);                                  // the other half of the head's channels
      e = valid ? fmaf(wl0, x0, fmaf(wl1, x1, e)) : -INFINITY;     // log2-domain score of (edge slot, head)
      const size_t u = static_cast<size_t>(ce0 + base + es);
      if (a_save != nullptr && valid && half == 0) a_save[u * NH + k] = e;   // raw score, normalised below
      const float mn = fmaxf(m, slots_max(e));
      const float sc = __builtin_amdgcn_exp2f(m - mn);             // exp2(-inf) = 0 on the first pass
      const float p = valid ? __builtin_amdgcn_exp2f(e - mn) : 0.f;
      den = fmaf(den, sc, slots_sum(p));
      s0 = fmaf(s0, sc, slots_sum(p * x0));
      s1 = fmaf(s1, sc, slots_sum(p * x1));
      m = mn;
    }
    const float inv = __builtin_amdgcn_rcpf(den);
    if ((lane & 15) == 0) {
      sw[k * FS + 0] = s0 * inv;
      sw[k * FS + 1] = s1 * inv;
    }
    if (a_save != nullptr && half == 0) {
      for (int base = 0; base < cdeg; base += kSlots) {
        if (base + es < cdeg) {
          float* ap = a_save + static_cast<size_t>(ce0 + base + es) * NH + k;
          *ap = __builtin_amdgcn_exp2f(*ap - m) * inv;
        }
      }
    }
    wave_sync_lds();
    // ---- epilogue: lane <-> channel ----------------------------------------------------------------------------
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int n = lane + kWave * jj;
      if (n < H) {
        const int kk = n / D;
        const float agg = fmaf(ws0[jj], sw[kk * FS + 0], fmaf(ws1[jj], sw[kk * FS + 1], bs[jj]));
        orow[n] = fmaxf(agg + res[jj], 0.f);
      }
    }
    wave_sync_lds();
  };

  // 64 destinations' meta data by vector loads (lane <-> destination), handed over with v_readlane
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
}

template <int D>
int launch_small(const float* x_src, const float* x_dst, const int32_t* seg_off, const int32_t* dst_order, int N,
                 const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                 const float* W_r, const float* b_r, float slope, float* out, int ld_out, float* a_save,
                 hipStream_t st) {
  const int grid = capped_grid(N, kWavesPerBlock, 512);   // persistent: 2 workgroups per CU, constants loaded once
  hipLaunchKernelGGL((gatv2_fwd_small_kernel<D>), dim3(grid), dim3(kThreads), 0, st, x_src, x_dst, seg_off, dst_order,
                     N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope, out, ld_out, a_save);
  return launch_status();
}

}  // namespace

int gatv2_fwd_small(int F_src, int nh, int D, const float* x_src, int E, const float* x_dst, const int32_t* seg_off,
                    const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d,
                    const float* b_d, const float* attn, const float* W_r, const float* b_r, float slope, float* out,
                    int ld_out, float* a_save, hipStream_t st) {
  if (F_src != 2 || nh != NH || static_cast<long long>(E) > 8LL * N) return UAVGNN_EUNSUPPORTED;
  if (D == 64) return launch_small<64>(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope,
                                       out, ld_out, a_save, st);
  if (D == 32) return launch_small<32>(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope,
                                       out, ld_out, a_save, st);
  if (D == 16) return launch_small<16>(x_src, x_dst, seg_off, dst_order, N, W_s, b_s, W_d, b_d, attn, W_r, b_r, slope,
                                       out, ld_out, a_save, st);
  return UAVGNN_EUNSUPPORTED;
}

}  // namespace uavgnn
