// K3b: targeted attention over the agent->agent `talk` relation (forward / backward).
//
// Replaces the DGL sequence at /root/reference/algos/madrqn/agents/gnn_agents.py:261-267
//   apply_edges(u_dot_v('s','q')) ; / key_size ; edge_softmax ; update_all(u_mul_e('v','a'), sum)
// and, in uniform mode (s == q == NULL), the UDF reduce `mailbox.mean(1)` of BaseComm / CommNet (:130-133,:214-216).
//
// One wavefront per destination (CSC).  Scores: lane <-> in-edge (each lane dots its source's signature row with the
// destination's query, staged once in LDS).  Softmax: wave reductions.  Aggregate: lane <-> message channel, the
// edge's weight and source id broadcast by __shfl, the source's value row read coalesced (256 B for msg = 64).
// Backward is two gather passes, no atomics: pass 1 per destination (d_q, and de per edge into scratch), pass 2 per
// source over the transposed CSC (d_s, d_v).
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int kMaxK = 64;
constexpr int kMaxMJ = 4;  // M <= 256

__global__ __launch_bounds__(kThreads) void talk_attn_fwd_kernel(
    const float* __restrict__ s, int ld_s, const float* __restrict__ q, int ld_q, const float* __restrict__ v,
    int ld_v, int K, int M, const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src, int N,
    float scale, float* __restrict__ c, int ld_c, float* __restrict__ a_save, const float* __restrict__ x_copy,
    int ld_x, int n_copy) {
  __shared__ __attribute__((aligned(16))) float sQ[kWavesPerBlock][kMaxK];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool uniform = (s == nullptr);
  const bool vec4 = !uniform && (K % 4 == 0) && (ld_s % 4 == 0) && ((reinterpret_cast<uintptr_t>(s) & 15) == 0);
  float* __restrict__ qw = sQ[wave];

  for (int d = blockIdx.x * kWavesPerBlock + wave; d < N; d += gridDim.x * kWavesPerBlock) {
    const int e0 = talk_off[d];
    const int deg = talk_off[d + 1] - e0;
    if (x_copy != nullptr) {   // fill the x half of the [x || c] row while the attention of this row is in flight
      const float* __restrict__ xs = x_copy + static_cast<size_t>(d) * ld_x;
      float* __restrict__ xd = c + static_cast<size_t>(d) * ld_c - n_copy;
      for (int i = lane; i < n_copy; i += kWave) xd[i] = xs[i];
    }
    float acc[kMaxMJ] = {0.f, 0.f, 0.f, 0.f};
    if (deg > 0) {
      float mx = 0.f, inv = 1.f / static_cast<float>(deg);
      float e_first = 0.f;    // score of edge `lane` of the first pass: stays in a register (in-degree <= 64 is the norm)
      if (!uniform) {
        if (lane < K) qw[lane] = q[static_cast<size_t>(d) * ld_q + lane];
        wave_sync_lds();
        // pass 1: raw scores, lane-local online max / sum
        float m = -INFINITY, den = 0.f;
        for (int base = 0; base < deg; base += kWave) {
          if (base + lane < deg) {
            const int u = talk_src[e0 + base + lane];
            const float* __restrict__ sr = s + static_cast<size_t>(u) * ld_s;
            float e = 0.f;
            if (vec4) {
              for (int kk = 0; kk < K; kk += 4) {
                const float4 sv = *reinterpret_cast<const float4*>(sr + kk);
                const float4 qv = *reinterpret_cast<const float4*>(qw + kk);
                e = fmaf(sv.x, qv.x, fmaf(sv.y, qv.y, fmaf(sv.z, qv.z, fmaf(sv.w, qv.w, e))));
              }
            } else {
              for (int kk = 0; kk < K; ++kk) e = fmaf(sr[kk], qw[kk], e);
            }
            e *= scale;
            if (base == 0) e_first = e;
            else a_save[e0 + base + lane] = e;          // only in-degrees > 64 round-trip through memory
            const float mn = fmaxf(m, e);
            den = fmaf(den, expf(m - mn), expf(e - mn));
            m = mn;
          }
        }
        mx = wave_max(m);
        const float dn = wave_sum(m == -INFINITY ? 0.f : den * expf(m - mx));
        inv = 1.f / dn;
        wave_sync_lds();
      }
      // pass 2: normalise, aggregate the value rows
      for (int base = 0; base < deg; base += kWave) {
        const bool valid = base + lane < deg;
        int u = 0;
        float a = 0.f;
        if (valid) {
          u = talk_src[e0 + base + lane];
          a = uniform ? inv : expf((base == 0 ? e_first : a_save[e0 + base + lane]) - mx) * inv;
          a_save[e0 + base + lane] = a;
        }
        const int cnt = min(kWave, deg - base);
        for (int i = 0; i < cnt; ++i) {
          const float ai = __shfl(a, i);
          const int ui = __shfl(u, i);
          const float* __restrict__ vr = v + static_cast<size_t>(ui) * ld_v;
#pragma unroll
          for (int jj = 0; jj < kMaxMJ; ++jj) {
            const int ch = lane + kWave * jj;
            if (ch < M) acc[jj] = fmaf(ai, vr[ch], acc[jj]);
          }
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < kMaxMJ; ++jj) {
      const int ch = lane + kWave * jj;
      if (ch < M) c[static_cast<size_t>(d) * ld_c + ch] = acc[jj];  // zero when the node has no in-edge
    }
  }
}

// pass 1 of the backward: per destination.  de[e] = a_e (da_e - sum_e' a_e' da_e') * scale, d_q[d] = sum_e de_e s[src_e]
__global__ __launch_bounds__(kThreads) void talk_attn_bwd_dst_kernel(
    const float* __restrict__ s, int ld_s, const float* __restrict__ v, int ld_v, int K, int M,
    const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src, int N, float scale,
    const float* __restrict__ a_save, const float* __restrict__ d_c, int ld_dc, float* __restrict__ d_q, int ld_dq,
    float* __restrict__ de_tmp) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int d = blockIdx.x * kWavesPerBlock + wave; d < N; d += gridDim.x * kWavesPerBlock) {
    const int e0 = talk_off[d];
    const int deg = talk_off[d + 1] - e0;
    // da_e = <d_c[d], v[src_e]> for every in-edge; T = sum_e a_e da_e.  Edges are taken 8 at a time: lane = (edge slot
    // l >> 3, channel phase l & 7) computes a strided partial dot, 3 shuffle steps finish all 8 dots together.
    float T = 0.f;
    const float* __restrict__ dcr = d_c + static_cast<size_t>(d) * ld_dc;
    for (int base = 0; base < deg; base += kWave) {
      const bool valid = base + lane < deg;
      const int u = valid ? talk_src[e0 + base + lane] : 0;
      const float a = valid ? a_save[e0 + base + lane] : 0.f;
      float da = 0.f;
      const int cnt = min(kWave, deg - base);
      for (int i0 = 0; i0 < cnt; i0 += 8) {
        const int slot = lane >> 3, ph = lane & 7;
        const int ui = __shfl(u, min(i0 + slot, kWave - 1));
        float p = 0.f;
        if (i0 + slot < cnt) {
          const float* __restrict__ vr = v + static_cast<size_t>(ui) * ld_v;
          for (int ch = ph; ch < M; ch += 8) p = fmaf(dcr[ch], vr[ch], p);
        }
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        p += __shfl_xor(p, 4);                         // every lane of slot k now holds da of edge i0 + k
        const float mine = __shfl(p, (lane & 7) << 3);  // lane i0 + k (k = lane & 7 when lane is in this group of 8)
        if (lane >= i0 && lane < i0 + 8) da = mine;
      }
      if (valid) de_tmp[e0 + base + lane] = da;  // raw da, finished below
      T += a * da;
    }
    T = wave_sum(T);
    float dq = 0.f;
    for (int base = 0; base < deg; base += kWave) {
      const bool valid = base + lane < deg;
      int u = 0;
      float de = 0.f;
      if (valid) {
        u = talk_src[e0 + base + lane];
        de = a_save[e0 + base + lane] * (de_tmp[e0 + base + lane] - T) * scale;
        de_tmp[e0 + base + lane] = de;
      }
      const int cnt = min(kWave, deg - base);
      for (int i = 0; i < cnt; ++i) {
        const float dei = __shfl(de, i);
        const int ui = __shfl(u, i);
        if (lane < K) dq = fmaf(dei, s[static_cast<size_t>(ui) * ld_s + lane], dq);
      }
    }
    if (lane < K) d_q[static_cast<size_t>(d) * ld_dq + lane] = dq;
  }
}

// pass 2 of the backward: per source over the transposed CSC.
//   d_v[u] = sum_{e: src=u} a_e d_c[dst_e]      d_s[u] = sum_e de_e q[dst_e]
__global__ __launch_bounds__(kThreads) void talk_attn_bwd_src_kernel(
    const float* __restrict__ q, int ld_q, int K, int M, const int32_t* __restrict__ t_off,
    const int32_t* __restrict__ t_dst, const int32_t* __restrict__ t_pos, int N, const float* __restrict__ a_save,
    const float* __restrict__ de_tmp, const float* __restrict__ d_c, int ld_dc, float* __restrict__ d_s, int ld_ds,
    float* __restrict__ d_v, int ld_dv) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool uniform = (q == nullptr);
  for (int u = blockIdx.x * kWavesPerBlock + wave; u < N; u += gridDim.x * kWavesPerBlock) {
    const int t0 = t_off[u], t1 = t_off[u + 1];
    float dv[kMaxMJ] = {0.f, 0.f, 0.f, 0.f};
    float ds = 0.f;
    for (int t = t0; t < t1; ++t) {
      const int dst = t_dst[t];
      const int pos = t_pos[t];
      const float a = a_save[pos];
      const float* __restrict__ dcr = d_c + static_cast<size_t>(dst) * ld_dc;
#pragma unroll
      for (int jj = 0; jj < kMaxMJ; ++jj) {
        const int ch = lane + kWave * jj;
        if (ch < M) dv[jj] = fmaf(a, dcr[ch], dv[jj]);
      }
      if (!uniform && lane < K) ds = fmaf(de_tmp[pos], q[static_cast<size_t>(dst) * ld_q + lane], ds);
    }
#pragma unroll
    for (int jj = 0; jj < kMaxMJ; ++jj) {
      const int ch = lane + kWave * jj;
      if (ch < M) d_v[static_cast<size_t>(u) * ld_dv + ch] = dv[jj];
    }
    if (!uniform && d_s != nullptr && lane < K) d_s[static_cast<size_t>(u) * ld_ds + lane] = ds;
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_talk_attn_fwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v,
                                    int K, int M, const int32_t* talk_off, const int32_t* talk_src, int N, float scale,
                                    float* c, int ld_c, float* a_save, const float* x_copy, int ld_x, int n_copy,
                                    uavgnn_stream_t stream) {
  if (N < 0 || !v || !talk_off || !c || !a_save || ((s == nullptr) != (q == nullptr))) return UAVGNN_EINVAL;
  if (M < 1 || M > kWave * kMaxMJ || (s && (K < 1 || K > kMaxK))) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  hipLaunchKernelGGL(talk_attn_fwd_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), s, ld_s, q, ld_q, v, ld_v, K, M, talk_off, talk_src, N, scale,
                     c, ld_c, a_save, x_copy, ld_x, n_copy);
  return launch_status();
}

extern "C" int uavgnn_talk_attn_bwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v,
                                    int K, int M, const int32_t* talk_off, const int32_t* talk_src,
                                    const int32_t* t_off, const int32_t* t_dst, const int32_t* t_pos, int N,
                                    float scale, const float* a_save, const float* d_c, int ld_dc, float* d_s,
                                    int ld_ds, float* d_q, int ld_dq, float* d_v, int ld_dv, float* de_tmp,
                                    uavgnn_stream_t stream) {
  if (N < 0 || !v || !talk_off || !t_off || !a_save || !d_c || !d_v || ((s == nullptr) != (q == nullptr)))
    return UAVGNN_EINVAL;
  if (s && (!d_s || !d_q || !de_tmp)) return UAVGNN_EINVAL;
  if (M < 1 || M > kWave * kMaxMJ || (s && (K < 1 || K > kMaxK))) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = capped_grid(N, kWavesPerBlock, 4096);
  if (s) {
    hipLaunchKernelGGL(talk_attn_bwd_dst_kernel, dim3(grid), dim3(kThreads), 0, st, s, ld_s, v, ld_v, K, M, talk_off,
                       talk_src, N, scale, a_save, d_c, ld_dc, d_q, ld_dq, de_tmp);
    int rc = launch_status();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(talk_attn_bwd_src_kernel, dim3(grid), dim3(kThreads), 0, st, q, ld_q, K, M, t_off, t_dst, t_pos,
                     N, a_save, de_tmp, d_c, ld_dc, d_s, ld_ds, d_v, ld_dv);
  return launch_status();
}
