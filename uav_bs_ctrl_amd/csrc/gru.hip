// K4 (pointwise half): GRU gate math fused into one pass over [N, 3H] pre-activations.
// Replaces the pointwise tail of nn.GRUCell used at /root/reference/algos/madrqn/agents/gnn_agents.py:29,:123,:164,
// :208,:246,:282 (PyTorch gate order r,z,n):
//   r = sigma(gi_r + gh_r)  z = sigma(gi_z + gh_z)  n = tanh(gi_n + r * gh_n)  h' = (1 - z) n + z h
// HBM-bound elementwise work: 16-byte accesses, grid-stride.  Backward recomputes the gates from gi/gh (cheaper than
// saving three more [N,H] tensors per BPTT step).
#include "common.h"

namespace uavgnn {
namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

template <int V>
struct Vec;
template <>
struct Vec<4> { using T = float4; };
template <>
struct Vec<1> { using T = float; };

template <int V>
__device__ __forceinline__ void ld(const float* p, float (&x)[V]) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
  } else {
    x[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void st(float* p, const float (&x)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
  } else {
    *p = x[0];
  }
}

template <int V>
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                            const float* __restrict__ h, long long total, int H,
                                                            float* __restrict__ h_out) {
  const int HV = H / V;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const long long row = i / HV;
    const int col = static_cast<int>(i - row * HV) * V;
    const float* gir = gi + row * 3 * H + col;
    const float* ghr = gh + row * 3 * H + col;
    float ir[V], iz[V], in_[V], hr[V], hz[V], hn[V], hh[V], o[V];
    ld<V>(gir, ir); ld<V>(gir + H, iz); ld<V>(gir + 2 * H, in_);
    ld<V>(ghr, hr); ld<V>(ghr + H, hz); ld<V>(ghr + 2 * H, hn);
    ld<V>(h + row * H + col, hh);
#pragma unroll
    for (int t = 0; t < V; ++t) {
      const float r = sigmoidf(ir[t] + hr[t]);
      const float z = sigmoidf(iz[t] + hz[t]);
      const float n = tanhf(fmaf(r, hn[t], in_[t]));
      o[t] = fmaf(z, hh[t] - n, n);  // (1-z) n + z h
    }
    st<V>(h_out + row * H + col, o);
  }
}

template <int V>
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                            const float* __restrict__ h,
                                                            const float* __restrict__ d_hout, long long total, int H,
                                                            float* __restrict__ d_gi, float* __restrict__ d_gh,
                                                            float* __restrict__ d_h) {
  const int HV = H / V;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const long long row = i / HV;
    const int col = static_cast<int>(i - row * HV) * V;
    const float* gir = gi + row * 3 * H + col;
    const float* ghr = gh + row * 3 * H + col;
    float ir[V], iz[V], in_[V], hr[V], hz[V], hn[V], hh[V], dho[V];
    ld<V>(gir, ir); ld<V>(gir + H, iz); ld<V>(gir + 2 * H, in_);
    ld<V>(ghr, hr); ld<V>(ghr + H, hz); ld<V>(ghr + 2 * H, hn);
    ld<V>(h + row * H + col, hh);
    ld<V>(d_hout + row * H + col, dho);
    float dr[V], dz[V], dni[V], dnh[V], dh[V];
#pragma unroll
    for (int t = 0; t < V; ++t) {
      const float r = sigmoidf(ir[t] + hr[t]);
      const float z = sigmoidf(iz[t] + hz[t]);
      const float n = tanhf(fmaf(r, hn[t], in_[t]));
      const float dn_pre = dho[t] * (1.f - z) * (1.f - n * n);
      dni[t] = dn_pre;
      dnh[t] = dn_pre * r;
      dr[t] = dn_pre * hn[t] * r * (1.f - r);
      dz[t] = dho[t] * (hh[t] - n) * z * (1.f - z);
      dh[t] = dho[t] * z;
    }
    float* dgir = d_gi + row * 3 * H + col;
    float* dghr = d_gh + row * 3 * H + col;
    st<V>(dgir, dr); st<V>(dgir + H, dz); st<V>(dgir + 2 * H, dni);
    st<V>(dghr, dr); st<V>(dghr + H, dz); st<V>(dghr + 2 * H, dnh);
    st<V>(d_h + row * H + col, dh);
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_gru_gates_fwd(const float* gi, const float* gh, const float* h, int N, int H, float* h_out,
                                    uavgnn_stream_t stream) {
  if (N < 0 || H <= 0 || !gi || !gh || !h || !h_out) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (H % 4 == 0) {
    const long long total = static_cast<long long>(N) * (H / 4);
    hipLaunchKernelGGL(gru_gates_fwd_kernel<4>, dim3(capped_grid(total, 256)), dim3(256), 0, s, gi, gh, h, total, H,
                       h_out);
  } else {
    const long long total = static_cast<long long>(N) * H;
    hipLaunchKernelGGL(gru_gates_fwd_kernel<1>, dim3(capped_grid(total, 256)), dim3(256), 0, s, gi, gh, h, total, H,
                       h_out);
  }
  return launch_status();
}

extern "C" int uavgnn_gru_gates_bwd(const float* gi, const float* gh, const float* h, const float* d_hout, int N,
                                    int H, float* d_gi, float* d_gh, float* d_h, uavgnn_stream_t stream) {
  if (N < 0 || H <= 0 || !gi || !gh || !h || !d_hout || !d_gi || !d_gh || !d_h) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (H % 4 == 0) {
    const long long total = static_cast<long long>(N) * (H / 4);
    hipLaunchKernelGGL(gru_gates_bwd_kernel<4>, dim3(capped_grid(total, 256)), dim3(256), 0, s, gi, gh, h, d_hout,
                       total, H, d_gi, d_gh, d_h);
  } else {
    const long long total = static_cast<long long>(N) * H;
    hipLaunchKernelGGL(gru_gates_bwd_kernel<1>, dim3(capped_grid(total, 256)), dim3(256), 0, s, gi, gh, h, d_hout,
                       total, H, d_gi, d_gh, d_h);
  }
  return launch_status();
}
