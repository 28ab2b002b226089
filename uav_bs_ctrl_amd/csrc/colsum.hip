// Bias gradients of the recurrent step: column sums of a [N, C] gradient matrix, accumulated IN PLACE into row-blocked
// partials acc[S, C] (acc[s] += sum of the rows of block s).
//
// Replaces `db = dY.sum(0)` of the nn.Linear / nn.GRUCell backward reached from
// /root/reference/algos/madrqn/learner.py:157 (loss.backward()) for the layers at gnn_agents.py:43-46,:237-246, once per
// BPTT step.  The caller keeps one acc per parameter for the whole backward and folds the S partials once per update
// (ops.WeightGradSink), so a step costs one streaming pass per matrix - no temporary, no separate `+=` launch.  A
// (row block, column tile) cell has exactly one owner workgroup and a fixed summation order: deterministic.
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;

// C >= 64 (or strided narrow matrices).  grid (S, column tiles of 64 * V); wave w of the block takes rows
// lo + w, lo + w + 4, ...; a lane owns V consecutive columns.
// MASK (V = 4 only): x is the gradient of y = relu(.) - the summed value is x where y > 0 and 0 elsewhere, and that masked
// gradient is also written to `out` (the ReLU backward of gnn_agents.py:99-102 and the bias gradient of the Linear in front of it in
// ONE pass over the [N, C] gradient instead of an elementwise pass followed by a reduction pass).
// `x` and `out` carry no __restrict__: the MASK instantiation may run IN PLACE (out == x, same row stride) - every element is read
// and written by the same lane, all loads of an iteration are issued before its stores.
// RM (MASK, C = 256: the 64 lanes of a wavefront own ONE row): row_absmax[row] = max |.| over the row of `out` - the row scales of
// the f16x2 product d(K1 output) = out W_aggr behind this kernel (csrc/gemm_h2.hip).
template <int V, bool MASK = false, bool RM = false>
__global__ __launch_bounds__(kThreads) void colsum_wide_kernel(const float* x, long long ld, int N, int C,
                                                               int rows_per_block, float* __restrict__ acc,
                                                               const float* __restrict__ y = nullptr, long long ldy = 0,
                                                               float* out = nullptr, long long ldo = 0,
                                                               float* __restrict__ row_absmax = nullptr) {
  __shared__ float part[kWaves][kWave * V];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x;
  const int c = blockIdx.y * (kWave * V) + lane * V;
  const int lo = s * rows_per_block, hi = min(N, lo + rows_per_block);
  float a[V];
#pragma unroll
  for (int t = 0; t < V; ++t) a[t] = 0.f;
  if (c < C) {
    const float* p = x + c;
    int r = lo + wave;
    for (; r + 3 * kWaves < hi; r += 4 * kWaves) {   // four rows in flight per lane
      float v[4][V];
      if constexpr (V == 4 && MASK) {
        float4 t4[4], y4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long row = r + u * kWaves;
          t4[u] = *reinterpret_cast<const float4*>(p + row * ld);
          y4[u] = *reinterpret_cast<const float4*>(y + row * ldy + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long row = r + u * kWaves;
          t4[u] = make_float4(y4[u].x > 0.f ? t4[u].x : 0.f, y4[u].y > 0.f ? t4[u].y : 0.f, y4[u].z > 0.f ? t4[u].z : 0.f,
                              y4[u].w > 0.f ? t4[u].w : 0.f);
          *reinterpret_cast<float4*>(out + row * ldo + c) = t4[u];
          v[u][0] = t4[u].x; v[u][1] = t4[u].y; v[u][2] = t4[u].z; v[u][3] = t4[u].w;
          if constexpr (RM) {
            const float m = wave_max(fmaxf(fmaxf(fabsf(t4[u].x), fabsf(t4[u].y)), fmaxf(fabsf(t4[u].z), fabsf(t4[u].w))));
            if (lane == 0) row_absmax[row] = m;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* q = p + static_cast<long long>(r + u * kWaves) * ld;
          if constexpr (V == 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(q);
            v[u][0] = t4.x; v[u][1] = t4.y; v[u][2] = t4.z; v[u][3] = t4.w;
          } else {
            v[u][0] = *q;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < V; ++t) a[t] += v[u][t];
    }
    for (; r < hi; r += kWaves) {
      const float* q = p + static_cast<long long>(r) * ld;
      float rmx = 0.f;
#pragma unroll
      for (int t = 0; t < V; ++t) {
        float v = q[t];
        if constexpr (MASK) {
          v = y[static_cast<long long>(r) * ldy + c + t] > 0.f ? v : 0.f;
          out[static_cast<long long>(r) * ldo + c + t] = v;
        }
        a[t] += v;
        if constexpr (RM) rmx = fmaxf(rmx, fabsf(v));
      }
      if constexpr (RM) {
        rmx = wave_max(rmx);
        if (lane == 0) row_absmax[r] = rmx;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < V; ++t) part[wave][lane * V + t] = a[t];
  __syncthreads();
  if (wave == 0 && c < C) {
#pragma unroll
    for (int t = 0; t < V; ++t) {
      float tot = part[0][lane * V + t];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) tot += part[w][lane * V + t];
      acc[static_cast<size_t>(s) * C + c + t] += tot;
    }
  }
}

// C < 64, rows contiguous (ld == C): the block's rows are one flat run of rows * C floats; thread t < L (L = the largest
// multiple of C <= 256) walks it with stride L, so it always sees column t % C.
__global__ __launch_bounds__(kThreads) void colsum_narrow_kernel(const float* __restrict__ x, int N, int C,
                                                                 int rows_per_block, float* __restrict__ acc) {
  __shared__ float part[kThreads];
  const int t = threadIdx.x;
  const int L = (kThreads / C) * C;
  const int s = blockIdx.x;
  const int lo = s * rows_per_block, hi = min(N, lo + rows_per_block);
  const float* __restrict__ p = x + static_cast<size_t>(lo) * C;
  const int total = (hi > lo ? hi - lo : 0) * C;
  float a = 0.f;
  if (t < L)
    for (int i = t; i < total; i += L) a += p[i];
  part[t] = a;
  __syncthreads();
  if (t < C) {
    float tot = 0.f;
    for (int g = t; g < L; g += C) tot += part[g];
    acc[static_cast<size_t>(s) * C + t] += tot;
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_colsum_acc(const float* x, long long ld, int N, int C, float* acc, int S,
                                 uavgnn_stream_t stream) {
  if (N < 0 || C < 1 || S < 1 || !acc || (N > 0 && !x) || ld < C) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = (N + S - 1) / S;
  if (C < kWave && ld == C) {
    hipLaunchKernelGGL(colsum_narrow_kernel, dim3(S), dim3(kThreads), 0, st, x, N, C, rows, acc);
    return launch_status();
  }
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (v4)
    hipLaunchKernelGGL(colsum_wide_kernel<4>, dim3(S, (C + 4 * kWave - 1) / (4 * kWave)), dim3(kThreads), 0, st, x, ld,
                       N, C, rows, acc);
  else
    hipLaunchKernelGGL(colsum_wide_kernel<1>, dim3(S, (C + kWave - 1) / kWave), dim3(kThreads), 0, st, x, ld, N, C,
                       rows, acc);
  return launch_status();
}

// out [N, C] = dy where y > 0 else 0 (the backward of y = relu(.)), acc[S, C] += its row-blocked column sums: one pass.  C % 4 == 0,
// row strides multiples of 4 floats, 16-byte aligned operands (UAVGNN_EUNSUPPORTED otherwise).  `out` may be `dy` itself (same pointer AND same row stride: in place); any other
// overlap of `out` with `dy`, and any overlap with `y`, is UAVGNN_EINVAL.
extern "C" int uavgnn_relu_bwd_colsum(const float* dy, long long ld, const float* y, long long ldy, float* out, long long ldo, int N,
                                      int C, float* acc, int S, uavgnn_stream_t stream) {
  if (N < 0 || C < 1 || S < 1 || !acc || (N > 0 && (!dy || !y || !out)) || ld < C || ldy < C || ldo < C) return UAVGNN_EINVAL;
  if (N > 0) {
    const uintptr_t o0 = reinterpret_cast<uintptr_t>(out), o1 = o0 + 4ull * (static_cast<uintptr_t>(N - 1) * ldo + C);
    const uintptr_t d0 = reinterpret_cast<uintptr_t>(dy), d1 = d0 + 4ull * (static_cast<uintptr_t>(N - 1) * ld + C);
    const uintptr_t y0 = reinterpret_cast<uintptr_t>(y), y1 = y0 + 4ull * (static_cast<uintptr_t>(N - 1) * ldy + C);
    const bool in_place = (o0 == d0 && ldo == ld);
    if ((!in_place && o0 < d1 && d0 < o1) || (o0 < y1 && y0 < o1)) return UAVGNN_EINVAL;
  }
  if ((C & 3) || (ld & 3) || (ldy & 3) || (ldo & 3) ||
      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const int rows = (N + S - 1) / S;
  hipLaunchKernelGGL((colsum_wide_kernel<4, true>), dim3(S, (C + 4 * kWave - 1) / (4 * kWave)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), dy, ld, N, C, rows, acc, y, ldy, out, ldo);
  return launch_status();
}

// ... that ALSO writes row_absmax [N] = max |.| over the rows of `out` (C = 256: one wavefront per row; UAVGNN_EUNSUPPORTED otherwise):
// the row scales of the f16x2 input-gradient product behind it (uavgnn_gemm_nt_h2).  Not in place (out != dy).
extern "C" int uavgnn_relu_bwd_colsum_rowmax(const float* dy, long long ld, const float* y, long long ldy, float* out, long long ldo, int N,
                                             int C, float* acc, int S, float* row_absmax, uavgnn_stream_t stream) {
  if (N < 0 || C < 1 || S < 1 || !acc || !row_absmax || (N > 0 && (!dy || !y || !out)) || ld < C || ldy < C || ldo < C || out == dy)
    return UAVGNN_EINVAL;
  if (C != 4 * kWave || (ld & 3) || (ldy & 3) || (ldo & 3) ||
      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const int rows = (N + S - 1) / S;
  hipLaunchKernelGGL((colsum_wide_kernel<4, true, true>), dim3(S, 1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), dy, ld, N, C,
                     rows, acc, y, ldy, out, ldo, row_absmax);
  return launch_status();
}
