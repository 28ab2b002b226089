// K3b, per-graph formulation: targeted attention over `talk` for a batch of SMALL graphs (one wavefront per graph).
//
// Same arithmetic as talk_attn.hip (gnn_agents.py:261-267; uniform mode :130-133,:214-216) for the shape the
// reference actually produces: dgl.batch of per-environment graphs with n_agents <= 16 whose talk edges never leave
// their environment (env_wrappers.py:139-154, common.py:45).  The per-destination kernel keeps 7 of 64 lanes busy in
// its score pass and chains four dependent global loads per destination.  Here a wavefront stages the projections of
// its graph's agents in LDS once, computes scores / softmax with lane <-> edge, scatters the weights into a dense
// n x n attention matrix A (LDS) and then both directions are tiny dense products with lane <-> channel and the
// operand rows held in registers:
//     forward   c[d]   = sum_u A[d][u] v[u]
//     backward  d_v[u] = sum_d A[d][u] d_c[d],   d_q[d] = sum_u dE[d][u] s[u],   d_s[u] = sum_d dE[d][u] q[d]
// so the backward needs neither the transposed CSC nor a scratch array, and no loop has an LDS load whose address
// depends on another LDS load.  Parallel edges add up in A (LDS atomics), as they do in the edge-wise formulation.
//
// Preconditions (checked per graph; a violating graph gets NaN outputs instead of silent corruption): agents of a
// graph <= n_max <= 16, edges of a graph <= NMAX^2 (NMAX = 8 or 16), sources inside the graph.  Shapes outside
// uavgnn_talk_attn_env_supported() (LDS budget) run on the per-destination kernels of talk_attn.hip.
#include <math.h>

#include "common.h"

namespace uavgnn {
namespace {

constexpr int kEnvMaxAgents = 16;
constexpr int kEnvMaxWaves = 4;
constexpr int kMaxK = 64;
constexpr int kMaxM = 256;

struct EnvDims {
  int nmax, emax, ldp, ldm;
};

__host__ __device__ inline EnvDims env_dims(int nmax, int M, int K) {
  EnvDims d;
  d.nmax = nmax;
  d.emax = nmax * nmax;
  d.ldp = (M + 2 * K) | 1;   // odd row strides: rows land in different LDS banks
  d.ldm = M | 1;
  return d;
}

// words of LDS per wavefront: P | OFF | SC | AD | SRC | DST | AV        (forward)
//                             P | DC | OFF | A | DA | AD | DE | SRC | DST | DV   (backward)
inline int fwd_words(const EnvDims& d) { return d.nmax * d.ldp + (d.nmax + 1) + 5 * d.emax; }
inline int bwd_words(const EnvDims& d) { return d.nmax * d.ldp + d.nmax * d.ldm + (d.nmax + 1) + 7 * d.emax; }
inline int waves_for(int words) {
  const int per_block = 64 * 1024 / 4;   // default dynamic-LDS ceiling of a workgroup
  int w = per_block / words;
  return w > kEnvMaxWaves ? kEnvMaxWaves : w;
}
inline int pick_nmax(int n_max) { return n_max <= 8 ? 8 : 16; }

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

// local source / destination of every edge of the graph (OFF already in LDS).  Returns true (wave-uniform) when some
// edge's source lies outside the graph's agents [a0, a0 + n): graph_off then does not delimit the talk relation (e.g. a
// container assembled by hand with the offsets of another relation) and the caller poisons the graph's output with NaN
// - a wrong precondition must fail loudly, never fall back to a self loop.
__device__ __forceinline__ bool stage_edges(const EnvGraph& g, int lane, const int32_t* __restrict__ talk_src,
                                            int first_src, const int* __restrict__ OFF, int* __restrict__ SRC,
                                            int* __restrict__ DST) {
  bool bad = false;
  for (int e = lane; e < g.E; e += kWave) {
    const int u = (e < kWave ? first_src : talk_src[g.e_lo + e]) - g.a0;
    bad |= (u < 0) | (u >= g.n);
    SRC[e] = u < 0 ? 0 : (u >= g.n ? g.n - 1 : u);
    int d = 0;
#pragma unroll 4
    for (int j = 1; j < g.n; ++j) d += e >= OFF[j];
    DST[e] = d;
  }
  return __any(bad);
}

__device__ __forceinline__ void poison_rows(float* __restrict__ dst, int ld, int a0, int n, int W, int lane) {
#pragma unroll 1
  for (int i = 0; i < n; ++i)
    for (int ch = lane; ch < W; ch += kWave) dst[static_cast<size_t>(a0 + i) * ld + ch] = NAN;
}

// Deterministic scatter of per-edge values VAL (LDS) into the dense [NMAX x NMAX] matrix Mx (zero-initialised): parallel
// (duplicate) edges of one (destination, source) pair are summed IN CSC ORDER by the lane of the first of them - no float
// atomics, so the result does not depend on which lane got to the LDS first.  Simple graphs: one
// pass over the <= n earlier / later in-edges of the destination, plain store.
__device__ __forceinline__ void scatter_edges(float* __restrict__ Mx, int nmax, const float* __restrict__ VAL,
                                              const int* __restrict__ SRC, const int* __restrict__ DST,
                                              const int* __restrict__ OFF, int E, int lane) {
  for (int e = lane; e < E; e += kWave) {
    const int d = DST[e], sidx = SRC[e];
    const int j0 = OFF[d], j1 = OFF[d + 1];
    bool first = true;
    for (int jx = j0; jx < e; ++jx) first = first && (SRC[jx] != sidx);
    if (first) {
      float acc = VAL[e];
      for (int jx = e + 1; jx < j1; ++jx)
        if (SRC[jx] == sidx) acc += VAL[jx];
      Mx[d * nmax + sidx] = acc;
    }
  }
}

// out[r][ch] = sum_{t < NMAX} W[r, t] * X[t][ch] for r < n, one 64-channel block at a time, X rows in registers
// (rows >= n are clamped copies multiplied by the zero padding of W).  TRANS: W[r, t] = Wm[t * NMAX + r], else
// Wm[r * NMAX + t].  The Wm reads are wave-uniform (LDS broadcast) and independent of each other.
template <int NMAX, bool TRANS>
__device__ __forceinline__ void dense_rows(const float* __restrict__ Wm, const float* __restrict__ X, int ldx, int n,
                                           int M, int lane, float* __restrict__ out, int ld_out) {
#pragma unroll 1
  for (int c0 = 0; c0 < M; c0 += kWave) {
    const int ch = c0 + lane, chc = ch < M ? ch : M - 1;
    float x[NMAX];
#pragma unroll
    for (int t = 0; t < NMAX; ++t) x[t] = X[(t < n ? t : n - 1) * ldx + chc];
#pragma unroll 2
    for (int r = 0; r < n; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < NMAX; ++t) acc = fmaf(TRANS ? Wm[t * NMAX + r] : Wm[r * NMAX + t], x[t], acc);
      if (ch < M) out[static_cast<size_t>(r) * ld_out + ch] = acc;
    }
  }
}

template <int NMAX>
__global__ __launch_bounds__(kEnvMaxWaves* kWave, 4) void talk_attn_env_fwd_kernel(
    const float* __restrict__ s, int ld_s, const float* __restrict__ q, int ld_q, const float* __restrict__ v,
    int ld_v, int K, int M, const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src,
    const int32_t* __restrict__ graph_off, int B, float scale, float* __restrict__ c, int ld_c,
    float* __restrict__ a_save, const float* __restrict__ x_copy, int ld_x, int n_copy, int words_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  const bool uniform = (s == nullptr);
  if (uniform) K = 0;
  const EnvDims dm = env_dims(NMAX, M, K);
  float* __restrict__ P = lds + wave * words_per_wave;
  int* __restrict__ OFF = reinterpret_cast<int*>(P + NMAX * dm.ldp);
  float* __restrict__ SC = reinterpret_cast<float*>(OFF + NMAX + 1);
  float* __restrict__ AD = SC + dm.emax;
  int* __restrict__ SRC = reinterpret_cast<int*>(AD + dm.emax);
  int* __restrict__ DST = SRC + dm.emax;
  float* __restrict__ AV = reinterpret_cast<float*>(DST + dm.emax);

  // blockIdx.y == 1: copy-only workgroups.  The x half of the [x || c] rows is a pure stream (2/3 of this kernel's
  // bytes); giving it its own wavefronts lets it overlap the LDS phases of the attention wavefronts, which otherwise
  // all load, compute and store in lock step (there is exactly one wavefront per graph in flight).
  if (blockIdx.y == 1) {
    const bool x4 = (n_copy % 4 == 0) && (ld_x % 4 == 0) && (ld_c % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(x_copy) & 15) == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0);
    for (int b = blockIdx.x * waves + wave; b < B; b += gridDim.x * waves) {
      const int a0 = graph_off[b], n = graph_off[b + 1] - a0;
#pragma unroll 1
      for (int i = 0; i < n; ++i) {
        const float* __restrict__ xs = x_copy + static_cast<size_t>(a0 + i) * ld_x;
        float* __restrict__ xd = c + static_cast<size_t>(a0 + i) * ld_c - n_copy;
        if (x4) {
#pragma unroll 4
          for (int k = lane * 4; k < n_copy; k += kWave * 4)
            *reinterpret_cast<float4*>(xd + k) = *reinterpret_cast<const float4*>(xs + k);
        } else {
          for (int k = lane; k < n_copy; k += kWave) xd[k] = xs[k];
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
    const bool ok = g.n <= NMAX && g.E <= dm.emax;
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
    for (int i = lane; i < dm.emax; i += kWave) AD[i] = 0.f;
    if (lane <= g.n) OFF[lane] = toff - g.e_lo;
    wave_sync_lds();
    const bool foreign = stage_edges(g, lane, talk_src, first_src, OFF, SRC, DST);
    wave_sync_lds();
    if (foreign) {   // an edge enters from outside the graph: precondition violated, fail loudly
      poison_rows(c, ld_c, g.a0, g.n, M, lane);
      continue;
    }
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
      a_save[g.e_lo + e] = a;
      AV[e] = a;
    }
    wave_sync_lds();
    scatter_edges(AD, NMAX, AV, SRC, DST, OFF, g.E, lane);   // parallel edges add up, in CSC order
    wave_sync_lds();
    dense_rows<NMAX, false>(AD, P, dm.ldp, g.n, M, lane, c + static_cast<size_t>(g.a0) * ld_c, ld_c);
    wave_sync_lds();   // the next graph restages the same LDS
  }
}

template <int NMAX>
__global__ __launch_bounds__(kEnvMaxWaves* kWave, 4) void talk_attn_env_bwd_kernel(
    const float* __restrict__ s, int ld_s, const float* __restrict__ q, int ld_q, const float* __restrict__ v,
    int ld_v, int K, int M, const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src,
    const int32_t* __restrict__ graph_off, int B, float scale, const float* __restrict__ a_save,
    const float* __restrict__ d_c, int ld_dc, float* __restrict__ d_s, int ld_ds, float* __restrict__ d_q, int ld_dq,
    float* __restrict__ d_v, int ld_dv, int words_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  const bool uniform = (s == nullptr);
  if (uniform) K = 0;
  const EnvDims dm = env_dims(NMAX, M, K);
  float* __restrict__ P = lds + wave * words_per_wave;
  float* __restrict__ DC = P + NMAX * dm.ldp;
  int* __restrict__ OFF = reinterpret_cast<int*>(DC + NMAX * dm.ldm);
  float* __restrict__ A = reinterpret_cast<float*>(OFF + NMAX + 1);
  float* __restrict__ DA = A + dm.emax;
  float* __restrict__ AD = DA + dm.emax;
  float* __restrict__ DE = AD + dm.emax;
  int* __restrict__ SRC = reinterpret_cast<int*>(DE + dm.emax);
  int* __restrict__ DST = SRC + dm.emax;
  float* __restrict__ DV = reinterpret_cast<float*>(DST + dm.emax);

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
    const bool ok = g.n <= NMAX && g.E <= dm.emax;
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
    for (int i = lane; i < dm.emax; i += kWave) {
      AD[i] = 0.f;
      DE[i] = 0.f;
    }
    for (int e = lane; e < g.E; e += kWave) A[e] = e < kWave ? first_a : a_save[g.e_lo + e];
    if (lane <= g.n) OFF[lane] = toff - g.e_lo;
    wave_sync_lds();
    const bool foreign = stage_edges(g, lane, talk_src, first_src, OFF, SRC, DST);
    wave_sync_lds();
    if (foreign) {
      poison_rows(d_v, ld_dv, g.a0, g.n, M, lane);
      continue;
    }
    scatter_edges(AD, NMAX, A, SRC, DST, OFF, g.E, lane);
    for (int e = lane; e < g.E; e += kWave) {
      if (!uniform) {   // da_e = <d_c[dst_e], v[src_e]>
        const float* __restrict__ dr = DC + DST[e] * dm.ldm;
        const float* __restrict__ vr = P + SRC[e] * dm.ldp;
        float acc = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < M; ++ch) acc = fmaf(dr[ch], vr[ch], acc);
        DA[e] = acc;
      }
    }
    wave_sync_lds();
    if (!uniform) {
      // dE[dst][src] += a_e (da_e - sum_{e' into dst} a_e' da_e') scale
      for (int e = lane; e < g.E; e += kWave) {
        const int d = DST[e];
        const int j1 = OFF[d + 1];
        float T = 0.f;
#pragma unroll 2
        for (int j = OFF[d]; j < j1; ++j) T = fmaf(A[j], DA[j], T);
        DV[e] = A[e] * (DA[e] - T) * scale;
      }
      wave_sync_lds();
      scatter_edges(DE, NMAX, DV, SRC, DST, OFF, g.E, lane);
      wave_sync_lds();
      // d_q[r][k] = sum_u dE[r][u] s[u][k],  d_s[r][k] = sum_d dE[d][r] q[d][k]:  lane = (row r, key column k)
      const int rpp = kWave / K;                      // rows per pass (K <= 64)
      const int lr = lane / K, k = lane - lr * K;
      for (int r0 = 0; r0 < g.n; r0 += rpp) {
        const int r = r0 + lr, rc = r < g.n ? r : g.n - 1;
        float acc_q = 0.f, acc_s = 0.f;
#pragma unroll
        for (int t = 0; t < NMAX; ++t) {
          const float* __restrict__ row = P + (t < g.n ? t : g.n - 1) * dm.ldp + M;
          acc_q = fmaf(DE[rc * NMAX + t], row[k], acc_q);
          acc_s = fmaf(DE[t * NMAX + rc], row[K + k], acc_s);
        }
        if (lr < rpp && r < g.n) {
          d_q[static_cast<size_t>(g.a0 + r) * ld_dq + k] = acc_q;
          if (d_s != nullptr) d_s[static_cast<size_t>(g.a0 + r) * ld_ds + k] = acc_s;
        }
      }
    }
    // d_v[u] = sum_d A[d][u] d_c[d]
    dense_rows<NMAX, true>(AD, DC, dm.ldm, g.n, M, lane, d_v + static_cast<size_t>(g.a0) * ld_dv, ld_dv);
    wave_sync_lds();
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_talk_attn_env_supported(int n_max, int M, int K) {
  if (n_max < 1 || n_max > kEnvMaxAgents || M < 1 || M > kMaxM || K < 0 || K > kMaxK) return 0;
  const EnvDims d = env_dims(pick_nmax(n_max), M, K);
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
  const int nmax = pick_nmax(n_max);
  const EnvDims d = env_dims(nmax, M, s ? K : 0);
  const int words = fwd_words(d);
  const int waves = waves_for(words);
  const dim3 grid(capped_grid(B, waves, 8192), x_copy ? 2 : 1), block(waves * kWave);
  const size_t lds_bytes = static_cast<size_t>(words) * waves * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nmax == 8)
    hipLaunchKernelGGL(talk_attn_env_fwd_kernel<8>, grid, block, lds_bytes, st, s, ld_s, q, ld_q, v, ld_v, K, M, talk_off,
                       talk_src, graph_off, B, scale, c, ld_c, a_save, x_copy, ld_x, n_copy, words);
  else
    hipLaunchKernelGGL(talk_attn_env_fwd_kernel<16>, grid, block, lds_bytes, st, s, ld_s, q, ld_q, v, ld_v, K, M,
                       talk_off, talk_src, graph_off, B, scale, c, ld_c, a_save, x_copy, ld_x, n_copy, words);
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
  const int nmax = pick_nmax(n_max);
  const EnvDims d = env_dims(nmax, M, s ? K : 0);
  const int words = bwd_words(d);
  const int waves = waves_for(words);
  const dim3 grid(capped_grid(B, waves, 8192)), block(waves * kWave);
  const size_t lds_bytes = static_cast<size_t>(words) * waves * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nmax == 8)
    hipLaunchKernelGGL(talk_attn_env_bwd_kernel<8>, grid, block, lds_bytes, st, s, ld_s, q, ld_q, v, ld_v, K, M, talk_off,
                       talk_src, graph_off, B, scale, a_save, d_c, ld_dc, d_s, ld_ds, d_q, ld_dq, d_v, ld_dv, words);
  else
    hipLaunchKernelGGL(talk_attn_env_bwd_kernel<16>, grid, block, lds_bytes, st, s, ld_s, q, ld_q, v, ld_v, K, M,
                       talk_off, talk_src, graph_off, B, scale, a_save, d_c, ld_dc, d_s, ld_ds, d_q, ld_dq, d_v, ld_dv,
                       words);
  return launch_status();
}
