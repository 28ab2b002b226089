// f1: device-side observation -> graph construction.
//
// Replaces the per-agent Python loops of /root/reference/algos/madrqn/utils/env_wrappers.py:65-89 (build_obs_graph: keep
// the rows whose visibility flag is 1, drop the flag column), :139-154 (build_comm_graph: edge i->j iff
// d_u2u[i,j] <= r_comm, self loops included) and dgl.batch / dgl.merge (:67,:137) for a whole batch of environments:
// padded observation tensors  gt [B,n,M,1+Fg]  ubs [B,n,U,1+Fu]  (column 0 = flag, mubs_cov.py:215-242) and the
// pairwise UBS distances d_u2u [B,n,n] go in, the segment layout of HeteroBatch comes out, without leaving the GPU.
// Pass 1 counts (one wavefront per agent row, ballot + popcount); the caller turns counts into offsets with a prefix
// sum; pass 2 compacts (rank of a kept row = popcount of lower lanes' ballot bits), so the order of the kept rows is
// the reference's (ascending m).  Pure byte/index work: results are bit-identical to the host builder.
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;

__device__ __forceinline__ int lanes_below(unsigned long long mask, int lane) {
  return __popcll(mask & ((1ull << lane) - 1ull));
}

// deg_seen[a], deg_near[a] for agent row a = b*n + i
__global__ __launch_bounds__(kThreads) void obs_degrees_kernel(const float* __restrict__ gt, int M, int Sg,
                                                               const float* __restrict__ ubs, int U, int Su, int N,
                                                               int32_t* __restrict__ deg_seen,
                                                               int32_t* __restrict__ deg_near) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int a = blockIdx.x * kWavesPerBlock + wave; a < N; a += gridDim.x * kWavesPerBlock) {
    int cs = 0, cn = 0;
    for (int m0 = 0; m0 < M; m0 += kWave) {
      const int m = m0 + lane;
      const bool keep = m < M && gt[(static_cast<size_t>(a) * M + m) * Sg] == 1.f;
      cs += __popcll(__ballot(keep));
    }
    for (int m0 = 0; m0 < U; m0 += kWave) {
      const int m = m0 + lane;
      const bool keep = m < U && ubs[(static_cast<size_t>(a) * U + m) * Su] == 1.f;
      cn += __popcll(__ballot(keep));
    }
    if (lane == 0) {
      deg_seen[a] = cs;
      deg_near[a] = cn;
    }
  }
}

template <int F>
__device__ __forceinline__ void compact_rows(const float* __restrict__ src, int rows, int S, size_t a, int lane,
                                             float* __restrict__ dst, int base) {
  for (int m0 = 0; m0 < rows; m0 += kWave) {
    const int m = m0 + lane;
    const float* r = src + (a * rows + m) * S;
    const bool keep = m < rows && r[0] == 1.f;
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      float* o = dst + static_cast<size_t>(base + lanes_below(mask, lane)) * F;
#pragma unroll
      for (int f = 0; f < F; ++f) o[f] = r[1 + f];
    }
    base += __popcll(mask);
  }
}

__global__ __launch_bounds__(kThreads) void obs_compact_kernel(const float* __restrict__ gt, int M, int Fg,
                                                               const float* __restrict__ ubs, int U, int Fu, int N,
                                                               const int32_t* __restrict__ seen_off,
                                                               const int32_t* __restrict__ near_off,
                                                               float* __restrict__ x_gt, float* __restrict__ x_ubs) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int a = blockIdx.x * kWavesPerBlock + wave; a < N; a += gridDim.x * kWavesPerBlock) {
    if (Fg == 4) compact_rows<4>(gt, M, 5, a, lane, x_gt, seen_off[a]);
    if (Fu == 2) compact_rows<2>(ubs, U, 3, a, lane, x_ubs, near_off[a]);
  }
}

// talk relation: one wavefront per environment, lane <-> destination j (n <= 64).
// deg_in[b*n+j] = #{i : d[b,i,j] <= r};  row_cnt[b] = #edges of env b (for the reference edge ids)
__global__ __launch_bounds__(kThreads) void talk_degrees_kernel(const float* __restrict__ d_u2u, int n, int B, float r,
                                                                int32_t* __restrict__ deg_in,
                                                                int32_t* __restrict__ env_edges) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int b = blockIdx.x * kWavesPerBlock + wave; b < B; b += gridDim.x * kWavesPerBlock) {
    const float* d = d_u2u + static_cast<size_t>(b) * n * n;
    int cnt = 0, tot = 0;
    for (int i = 0; i < n; ++i) {
      const bool e = lane < n && d[i * n + lane] <= r;
      cnt += e ? 1 : 0;
      tot += __popcll(__ballot(e));
    }
    if (lane < n) deg_in[b * n + lane] = cnt;
    if (lane == 0) env_edges[b] = tot;
  }
}

__global__ __launch_bounds__(kThreads) void talk_compact_kernel(const float* __restrict__ d_u2u, int n, int B, float r,
                                                                const int32_t* __restrict__ talk_off,
                                                                const int32_t* __restrict__ env_base,
                                                                int32_t* __restrict__ talk_src,
                                                                int32_t* __restrict__ talk_eid) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int b = blockIdx.x * kWavesPerBlock + wave; b < B; b += gridDim.x * kWavesPerBlock) {
    const float* d = d_u2u + static_cast<size_t>(b) * n * n;
    int pos = lane < n ? talk_off[b * n + lane] : 0;   // next free CSC slot of destination j = lane
    int eid = env_base[b];                             // reference edge ids run i-major inside an environment
    for (int i = 0; i < n; ++i) {
      const bool e = lane < n && d[i * n + lane] <= r;
      const unsigned long long mask = __ballot(e);
      if (e) {
        talk_src[pos] = b * n + i;
        talk_eid[pos] = eid + lanes_below(mask, lane);
        ++pos;
      }
      eid += __popcll(mask);
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_obs_degrees(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, int N,
                                  int32_t* deg_seen, int32_t* deg_near, uavgnn_stream_t stream) {
  if (N < 0 || (!gt && M > 0) || (!ubs && U > 0) || !deg_seen || !deg_near || M < 0 || U < 0) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(obs_degrees_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), gt, M, Fg + 1, ubs, U, Fu + 1, N, deg_seen, deg_near);
  return launch_status();
}

extern "C" int uavgnn_obs_compact(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, int N,
                                  const int32_t* seen_off, const int32_t* near_off, float* x_gt, float* x_ubs,
                                  uavgnn_stream_t stream) {
  if (N < 0 || (!gt && M > 0) || (!ubs && U > 0) || !seen_off || !near_off) return UAVGNN_EINVAL;
  if (Fg != 4 || Fu != 2) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  hipLaunchKernelGGL(obs_compact_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), gt, M, Fg, ubs, U, Fu, N, seen_off, near_off, x_gt, x_ubs);
  return launch_status();
}

extern "C" int uavgnn_talk_degrees(const float* d_u2u, int n, int B, float r_comm, int32_t* deg_in,
                                   int32_t* env_edges, uavgnn_stream_t stream) {
  if (B < 0 || !d_u2u || !deg_in || !env_edges) return UAVGNN_EINVAL;
  if (n < 1 || n > kWave) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  hipLaunchKernelGGL(talk_degrees_kernel, dim3(capped_grid(B, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), d_u2u, n, B, r_comm, deg_in, env_edges);
  return launch_status();
}

extern "C" int uavgnn_talk_compact(const float* d_u2u, int n, int B, float r_comm, const int32_t* talk_off,
                                   const int32_t* env_base, int32_t* talk_src, int32_t* talk_eid,
                                   uavgnn_stream_t stream) {
  if (B < 0 || !d_u2u || !talk_off || !env_base || !talk_src || !talk_eid) return UAVGNN_EINVAL;
  if (n < 1 || n > kWave) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  hipLaunchKernelGGL(talk_compact_kernel, dim3(capped_grid(B, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), d_u2u, n, B, r_comm, talk_off, env_base, talk_src, talk_eid);
  return launch_status();
}
