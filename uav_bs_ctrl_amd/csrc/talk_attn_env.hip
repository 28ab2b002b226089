// K3b, per-graph formulation: targeted attention over `talk` for a batch of SMALL graphs (one wavefront per graph).
//
// Same arithmetic as talk_attn.hip (gnn_agents.py:261-267; uniform mode :130-133,:214-216) for the shape the
// reference actually produces: dgl.batch of per-environment graphs with n_agents <= 16 whose talk edges never leave
// their environment (env_wrappers.py:139-154, common.py:45).  The per-destination kernel keeps 7 of 64 lanes busy in
// its score pass and chains four dependent global loads per destination; here a wavefront stages the projections of
// its graph's agents in LDS once (coalesced), then lane <-> edge for scores / softmax and lane <-> channel for the
// aggregation, all out of LDS.  The backward needs NO transpose of the CSC: gradients w.r.t. the sources are
// accumulated in LDS while walking the graph's edges in CSC order (same summation order as the transposed gather).
//
// Preconditions (checked per graph; a violating graph gets NaN outputs instead of silent corruption): agents of a
// graph <= n_max <= 16, edges of a graph <= n_max^2 (simple graph), sources inside the graph.  Shapes outside
// uavgnn_talk_attn_env_supported() (LDS budget) run on the per-destination kernels of talk_attn.hip.
#include <math.h>

#include "common.h"

namespace uavgnn {
namespace {

constexpr int kEnvMaxAgents = 16;
constexpr int kEnvMaxWaves = 4;
constexpr int kMaxK = 64;
constexpr int kMaxMJ = 4;   // M <= 256

struct EnvDims {
  int n_max, emax, ldp, ldm, ldk;
};

__host__ __device__ inline EnvDims env_dims(int n_max, int M, int K) {
  EnvDims d;
  d.n_max = n_max;
  d.emax = n_max * n_max;
  d.ldp = (M + 2 * K) | 1;   // odd row strides: rows land in different LDS banks
  d.ldm = M | 1;
  d.ldk = K | 1;
  return d;
}

inline int fwd_words(const EnvDims& d) { return d.n_max * d.ldp + (d.n_max + 1) + 4 * d.emax; }
inline int bwd_words(const EnvDims& d) {
  return d.n_max * d.ldp + 2 * d.n_max * d.ldm + d.n_max * d.ldk + (d.n_max + 1) + 5 * d.emax;
}
inline int waves_for(int words) {
  const int per_block = 64 * 1024 / 4;   // default dynamic-LDS ceiling of a workgroup
  int w = per_block / words;
  return w > kEnvMaxWaves ? kEnvMaxWaves : w;
}

struct EnvGraph {
  int a0, n, e_lo, E;
};

// Staging.  A wavefront owns one graph (a few KB), so what matters is the number of DEPENDENT global round trips,
// not bytes: the first 8 rows x first 64 columns of every matrix are loaded back to back (row index clamped instead
// of predicated: no branches) and only then written to LDS; whatever is left (more than 8 agents, rows wider than 64)
// goes through the plain loop of stage_rest.
constexpr int kRows = 8;

__device__ __forceinline__ void load_rows(const float* __restrict__ src, int ld, int a0, int n, int W, int lane,
                                          float (&r)[kRows]) {
  const float* __restrict__ base = src + static_cast<size_t>(a0) * ld;   // wave-uniform base, 32-bit lane offsets
  const int colc = lane < W ? lane : 0;
#pragma unroll
  for (int t = 0; t < kRows; ++t) r[t] = base[static_cast<unsigned>((t < n ? t : n - 1) * ld + colc)];
}

__device__ __forceinline__ void store_rows(float* __restrict__ dst, int lds_ld, int n, int W, int lane,
                                           const float (&r)[kRows]) {
#pragma unroll
  for (int t = 0; t < kRows; ++t)
    if (t < n && lane < W) dst[t * lds_ld + lane] = r[t];
}

__device__ __forceinline__ void stage_rest(const float* __restrict__ src, int ld, int a0, int n, int W,
                                           float* __restrict__ dst, int lds_ld, int lane) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const float* __restrict__ r = src + static_cast<size_t>(a0 + i) * ld;
#pragma unroll 1
    for (int k = (i < kRows ? kWave : 0) + lane; k < W; k += kWave) dst[i * lds_ld + k] = r[k];
  }
}

// local source / destination of every edge of the graph (OFF already in LDS)
__device__ __forceinline__ void stage_edges(const EnvGraph& g, int lane, const int32_t* __restrict__ talk_src,
                                            int first_src, const int* __restrict__ OFF, int* __restrict__ SRC,
                                            int* __restrict__ DST) {
  for (int e = lane; e < g.E; e += kWave) {
    const int u = (e < kWave ? first_src : talk_src[g.e_lo + e]) - g.a0;
    SRC[e] = u < 0 ? 0 : (u >= g.n ? g.n - 1 : u);
    int d = 0;
#pragma unroll 4
    for (int j = 1; j < g.n; ++j) d += e >= OFF[j];
    DST[e] = d;
  }
}

__global__ __launch_bounds__(kEnvMaxWaves* kWave, 4) void talk_attn_env_fwd_kernel(
    const float* __restrict__ s, int ld_s, const float* __restrict__ q, int ld_q, const float* __restrict__ v,
    int ld_v, int K, int M, const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src,
    const int32_t* __restrict__ graph_off, int B, int n_max, float scale, float* __restrict__ c, int ld_c,
    float* __restrict__ a_save, const float* __restrict__ x_copy, int ld_x, int n_copy, int words_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  const bool uniform = (s == nullptr);
  if (uniform) K = 0;
  const EnvDims dm = env_dims(n_max, M, K);
  float* __restrict__ P = lds + wave * words_per_wave;
  int* __restrict__ OFF = reinterpret_cast<int*>(P + dm.n_max * dm.ldp);
  float* __restrict__ SC = reinterpret_cast<float*>(OFF + dm.n_max + 1);
  float* __restrict__ AW = SC + dm.emax;
  int* __restrict__ SRC = reinterpret_cast<int*>(AW + dm.emax);
  int* __restrict__ DST = SRC + dm.emax;
  const bool x4 = x_copy != nullptr && (n_copy % 4 == 0) && (ld_x % 4 == 0) && (ld_c % 4 == 0) &&
                  ((reinterpret_cast<uintptr_t>(x_copy) & 15) == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0);

  // blockIdx.y == 1: copy-only workgroups.  The x half of the [x || c] rows is a pure stream (2/3 of this kernel's
  // bytes); giving it its own wavefronts lets it overlap the LDS phases of the attention wavefronts, which otherwise
  // all load, compute and store in lock step (there is exactly one wavefront per graph in flight).
  if (blockIdx.y == 1) {
    for (int b = blockIdx.x * waves + wave; b < B; b += gridDim.x * waves) {
      EnvGraph g;
      g.a0 = graph_off[b];
      g.n = graph_off[b + 1] - g.a0;
      if (x_copy != nullptr) {
#pragma unroll 1
        for (int i = 0; i < g.n; ++i) {
          const float* __restrict__ xs = x_copy + static_cast<size_t>(g.a0 + i) * ld_x;
          float* __restrict__ xd = c + static_cast<size_t>(g.a0 + i) * ld_c - n_copy;
          if (x4) {
#pragma unroll 4
            for (int k = lane * 4; k < n_copy; k += kWave * 4)
              *reinterpret_cast<float4*>(xd + k) = *reinterpret_cast<const float4*>(xs + k);
          } else {
            for (int k = lane; k < n_copy; k += kWave) xd[k] = xs[k];
          }
        }
      }
    }
    return;
  }

  for (int b = blockIdx.x * waves + wave; b < B; b += gridDim.x * waves) {
    EnvGraph g;
    g.a0 = graph_off[b];
    g.n = graph_off[b + 1] - g.a0;
    const int toff = lane <= g.n ? talk_off[g.a0 + lane] : 0;
    if (g.n == 0) continue;
    float rv[kRows], rs[kRows], rq[kRows];    // issued before anything waits on toff
    load_rows(v, ld_v, g.a0, g.n, M, lane, rv);
    if (!uniform) {
      load_rows(s, ld_s, g.a0, g.n, K, lane, rs);
      load_rows(q, ld_q, g.a0, g.n, K, lane, rq);
    }
    g.e_lo = __shfl(toff, 0);
    g.E = __shfl(toff, g.n < kWave ? g.n : kWave - 1) - g.e_lo;
    const bool ok = g.n <= dm.n_max && g.E <= dm.emax;
    const int first_src = (ok && lane < g.E) ? talk_src[g.e_lo + lane] : 0;
    if (!ok) {   // precondition violated: fail loudly
#pragma unroll 1
      for (int i = 0; i < g.n; ++i)
        for (int ch = lane; ch < M; ch += kWave) c[static_cast<size_t>(g.a0 + i) * ld_c + ch] = NAN;
      continue;
    }
    store_rows(P, dm.ldp, g.n, M, lane, rv);
    if (!uniform) {
      store_rows(P + M, dm.ldp, g.n, K, lane, rs);
      store_rows(P + M + K, dm.ldp, g.n, K, lane, rq);
    }
    if (g.n > kRows || M > kWave) {
      stage_rest(v, ld_v, g.a0, g.n, M, P, dm.ldp, lane);
      if (!uniform) {
        stage_rest(s, ld_s, g.a0, g.n, K, P + M, dm.ldp, lane);
        stage_rest(q, ld_q, g.a0, g.n, K, P + M + K, dm.ldp, lane);
      }
    }
    if (lane <= g.n) OFF[lane] = toff - g.e_lo;
    wave_sync_lds();
    stage_edges(g, lane, talk_src, first_src, OFF, SRC, DST);
    wave_sync_lds();
    if (!uniform) {
      for (int e = lane; e < g.E; e += kWave) {
        const float* __restrict__ sr = P + SRC[e] * dm.ldp + M;
        const float* __restrict__ qr = P + DST[e] * dm.ldp + M + K;
        float acc = 0.f;
#pragma unroll 4
        for (int k = 0; k < K; ++k) acc = fmaf(sr[k], qr[k], acc);
        SC[e] = acc * scale;
      }
      wave_sync_lds();
    }
    for (int e = lane; e < g.E; e += kWave) {   // softmax over the in-edges of the edge's destination
      const int d = DST[e];
      const int j0 = OFF[d], j1 = OFF[d + 1];
      float a;
      if (uniform) {
        a = 1.f / static_cast<float>(j1 - j0);
      } else {
        float m = -INFINITY;
#pragma unroll 2
        for (int j = j0; j < j1; ++j) m = fmaxf(m, SC[j]);
        float den = 0.f;
#pragma unroll 2
        for (int j = j0; j < j1; ++j) den += expf(SC[j] - m);
        a = expf(SC[e] - m) / den;
      }
      AW[e] = a;
      a_save[g.e_lo + e] = a;
    }
    wave_sync_lds();
#pragma unroll 1
    for (int d = 0; d < g.n; ++d) {
      float acc[kMaxMJ] = {0.f, 0.f, 0.f, 0.f};
      const int j1 = OFF[d + 1];
#pragma unroll 2
      for (int j = OFF[d]; j < j1; ++j) {
        const float aj = AW[j];
        const float* __restrict__ vr = P + SRC[j] * dm.ldp;
#pragma unroll
        for (int jj = 0; jj < kMaxMJ; ++jj) {
          const int ch = lane + kWave * jj;
          if (ch < M) acc[jj] = fmaf(aj, vr[ch], acc[jj]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < kMaxMJ; ++jj) {
        const int ch = lane + kWave * jj;
        if (ch < M) c[static_cast<size_t>(g.a0 + d) * ld_c + ch] = acc[jj];
      }
    }
    wave_sync_lds();   // the next graph restages the same LDS
  }
}

__global__ __launch_bounds__(kEnvMaxWaves* kWave, 4) void talk_attn_env_bwd_kernel(
    const float* __restrict__ s, int ld_s, const float* __restrict__ q, int ld_q, const float* __restrict__ v,
    int ld_v, int K, int M, const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src,
    const int32_t* __restrict__ graph_off, int B, int n_max, float scale, const float* __restrict__ a_save,
    const float* __restrict__ d_c, int ld_dc, float* __restrict__ d_s, int ld_ds, float* __restrict__ d_q, int ld_dq,
    float* __restrict__ d_v, int ld_dv, int words_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  const bool uniform = (s == nullptr);
  if (uniform) K = 0;
  const EnvDims dm = env_dims(n_max, M, K);
  float* __restrict__ P = lds + wave * words_per_wave;
  float* __restrict__ DC = P + dm.n_max * dm.ldp;
  float* __restrict__ DV = DC + dm.n_max * dm.ldm;
  float* __restrict__ DS = DV + dm.n_max * dm.ldm;
  int* __restrict__ OFF = reinterpret_cast<int*>(DS + dm.n_max * dm.ldk);
  float* __restrict__ A = reinterpret_cast<float*>(OFF + dm.n_max + 1);
  float* __restrict__ DA = A + dm.emax;
  float* __restrict__ DE = DA + dm.emax;
  int* __restrict__ SRC = reinterpret_cast<int*>(DE + dm.emax);
  int* __restrict__ DST = SRC + dm.emax;

  for (int b = blockIdx.x * waves + wave; b < B; b += gridDim.x * waves) {
    EnvGraph g;
    g.a0 = graph_off[b];
    g.n = graph_off[b + 1] - g.a0;
    const int toff = lane <= g.n ? talk_off[g.a0 + lane] : 0;
    if (g.n == 0) continue;
    float rd[kRows], rv[kRows], rs[kRows], rq[kRows];    // issued before anything waits on toff
    load_rows(d_c, ld_dc, g.a0, g.n, M, lane, rd);
    load_rows(v, ld_v, g.a0, g.n, M, lane, rv);
    if (!uniform) {
      load_rows(s, ld_s, g.a0, g.n, K, lane, rs);
      load_rows(q, ld_q, g.a0, g.n, K, lane, rq);
    }
    g.e_lo = __shfl(toff, 0);
    g.E = __shfl(toff, g.n < kWave ? g.n : kWave - 1) - g.e_lo;
    const bool ok = g.n <= dm.n_max && g.E <= dm.emax;
    const int first_src = (ok && lane < g.E) ? talk_src[g.e_lo + lane] : 0;
    const float first_a = (ok && lane < g.E) ? a_save[g.e_lo + lane] : 0.f;
    if (!ok) {
#pragma unroll 1
      for (int i = 0; i < g.n; ++i)
        for (int ch = lane; ch < M; ch += kWave) d_v[static_cast<size_t>(g.a0 + i) * ld_dv + ch] = NAN;
      continue;
    }
    store_rows(DC, dm.ldm, g.n, M, lane, rd);
    store_rows(P, dm.ldp, g.n, M, lane, rv);
    if (!uniform) {
      store_rows(P + M, dm.ldp, g.n, K, lane, rs);
      store_rows(P + M + K, dm.ldp, g.n, K, lane, rq);
    }
    if (g.n > kRows || M > kWave) {
      stage_rest(d_c, ld_dc, g.a0, g.n, M, DC, dm.ldm, lane);
      stage_rest(v, ld_v, g.a0, g.n, M, P, dm.ldp, lane);
      if (!uniform) {
        stage_rest(s, ld_s, g.a0, g.n, K, P + M, dm.ldp, lane);
        stage_rest(q, ld_q, g.a0, g.n, K, P + M + K, dm.ldp, lane);
      }
    }
#pragma unroll 1
    for (int i = 0; i < g.n; ++i) {
      for (int ch = lane; ch < M; ch += kWave) DV[i * dm.ldm + ch] = 0.f;
      if (!uniform && lane < K) DS[i * dm.ldk + lane] = 0.f;
    }
    for (int e = lane; e < g.E; e += kWave) A[e] = e < kWave ? first_a : a_save[g.e_lo + e];
    if (lane <= g.n) OFF[lane] = toff - g.e_lo;
    wave_sync_lds();
    stage_edges(g, lane, talk_src, first_src, OFF, SRC, DST);
    wave_sync_lds();
    if (!uniform) {
      // da_e = <d_c[dst_e], v[src_e]>
      for (int e = lane; e < g.E; e += kWave) {
        const float* __restrict__ dr = DC + DST[e] * dm.ldm;
        const float* __restrict__ vr = P + SRC[e] * dm.ldp;
        float acc = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < M; ++ch) acc = fmaf(dr[ch], vr[ch], acc);
        DA[e] = acc;
      }
      wave_sync_lds();
      // de_e = a_e (da_e - sum_{e' into dst} a_e' da_e') scale
      for (int e = lane; e < g.E; e += kWave) {
        const int d = DST[e];
        const int j1 = OFF[d + 1];
        float T = 0.f;
#pragma unroll 2
        for (int j = OFF[d]; j < j1; ++j) T = fmaf(A[j], DA[j], T);
        DE[e] = A[e] * (DA[e] - T) * scale;
      }
      wave_sync_lds();
      // d_q[d] = sum_{e into d} de_e s[src_e]
      for (int idx = lane; idx < g.n * K; idx += kWave) {
        const int d = idx / K, k = idx - d * K;
        const int j1 = OFF[d + 1];
        float acc = 0.f;
#pragma unroll 2
        for (int j = OFF[d]; j < j1; ++j) acc = fmaf(DE[j], P[SRC[j] * dm.ldp + M + k], acc);
        d_q[static_cast<size_t>(g.a0 + d) * ld_dq + k] = acc;
      }
    }
    // d_v[u] = sum_{e out of u} a_e d_c[dst_e],  d_s[u] = sum_{e out of u} de_e q[dst_e]: the graph's edges in CSC
    // order, accumulators in LDS (a lane owns its channel column: no conflicts, in-order LDS traffic per wave)
#pragma unroll 2
    for (int e = 0; e < g.E; ++e) {
      const int u = SRC[e], d = DST[e];
      const float a = A[e];
#pragma unroll
      for (int jj = 0; jj < kMaxMJ; ++jj) {
        const int ch = lane + kWave * jj;
        if (ch < M) DV[u * dm.ldm + ch] = fmaf(a, DC[d * dm.ldm + ch], DV[u * dm.ldm + ch]);
      }
      if (!uniform && lane < K)
        DS[u * dm.ldk + lane] = fmaf(DE[e], P[d * dm.ldp + M + K + lane], DS[u * dm.ldk + lane]);
    }
    wave_sync_lds();
#pragma unroll 1
    for (int i = 0; i < g.n; ++i) {
      for (int ch = lane; ch < M; ch += kWave)
        d_v[static_cast<size_t>(g.a0 + i) * ld_dv + ch] = DV[i * dm.ldm + ch];
      if (!uniform && d_s != nullptr && lane < K) d_s[static_cast<size_t>(g.a0 + i) * ld_ds + lane] = DS[i * dm.ldk + lane];
    }
    wave_sync_lds();
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_talk_attn_env_supported(int n_max, int M, int K) {
  if (n_max < 1 || n_max > kEnvMaxAgents || M < 1 || M > kWave * kMaxMJ || K < 0 || K > kMaxK) return 0;
  const EnvDims d = env_dims(n_max, M, K);
  return waves_for(bwd_words(d)) >= 1 ? 1 : 0;
}

extern "C" int uavgnn_talk_attn_env_fwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v,
                                        int K, int M, const int32_t* talk_off, const int32_t* talk_src,
                                        const int32_t* graph_off, int B, int n_max, float scale, float* c, int ld_c,
                                        float* a_save, const float* x_copy, int ld_x, int n_copy,
                                        uavgnn_stream_t stream) {
  if (B < 0 || !v || !talk_off || !graph_off || !c || !a_save || ((s == nullptr) != (q == nullptr)))
    return UAVGNN_EINVAL;
  if (!uavgnn_talk_attn_env_supported(n_max, M, s ? K : 0) || (s && K < 1)) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  const EnvDims d = env_dims(n_max, M, s ? K : 0);
  const int words = fwd_words(d);
  const int waves = waves_for(words);
  hipLaunchKernelGGL(talk_attn_env_fwd_kernel, dim3(capped_grid(B, waves, 8192), x_copy ? 2 : 1), dim3(waves * kWave),
                     static_cast<size_t>(words) * waves * sizeof(float), static_cast<hipStream_t>(stream), s, ld_s, q,
                     ld_q, v, ld_v, K, M, talk_off, talk_src, graph_off, B, n_max, scale, c, ld_c, a_save, x_copy,
                     ld_x, n_copy, words);
  return launch_status();
}

extern "C" int uavgnn_talk_attn_env_bwd(const float* s, int ld_s, const float* q, int ld_q, const float* v, int ld_v,
                                        int K, int M, const int32_t* talk_off, const int32_t* talk_src,
                                        const int32_t* graph_off, int B, int n_max, float scale, const float* a_save,
                                        const float* d_c, int ld_dc, float* d_s, int ld_ds, float* d_q, int ld_dq,
                                        float* d_v, int ld_dv, uavgnn_stream_t stream) {
  if (B < 0 || !v || !talk_off || !graph_off || !a_save || !d_c || !d_v || ((s == nullptr) != (q == nullptr)))
    return UAVGNN_EINVAL;
  if (s && (!d_s || !d_q)) return UAVGNN_EINVAL;
  if (!uavgnn_talk_attn_env_supported(n_max, M, s ? K : 0) || (s && K < 1)) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  const EnvDims d = env_dims(n_max, M, s ? K : 0);
  const int words = bwd_words(d);
  const int waves = waves_for(words);
  hipLaunchKernelGGL(talk_attn_env_bwd_kernel, dim3(capped_grid(B, waves, 8192)), dim3(waves * kWave),
                     static_cast<size_t>(words) * waves * sizeof(float), static_cast<hipStream_t>(stream), s, ld_s, q,
                     ld_q, v, ld_v, K, M, talk_off, talk_src, graph_off, B, n_max, scale, a_save, d_c, ld_dc, d_s,
                     ld_ds, d_q, ld_dq, d_v, ld_dv, words);
  return launch_status();
}
