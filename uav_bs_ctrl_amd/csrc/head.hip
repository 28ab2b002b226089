// The Q head of the agent - q = h' W_out^T + b_out, /root/reference/algos/madrqn/agents/gnn_agents.py:43-46,:56 (nn.Linear(H, n_actions)) -
// for n_actions <= 16: [N, 256] x [256, 9] at C3.  The vendor GEMM takes 11.8 us for it (a 16 x 256 macro tile, 82 TFLOP/s: neither
// compute- nor bandwidth-bound); what the layer costs is reading h' once - 33.5 MB, ~7 us.
//
// One wavefront per tile of 16 rows, fp32 products on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 multiply, fp32 accumulate, a
// fixed order): lane (i, g) loads float4 s of row i - columns 16 s + 4 g .. + 3, 64 contiguous bytes per row and instruction - and
// feeds its four components to four MFMAs as the A element of K slot g; lane (j, g) holds the same four columns of row j of W_out
// (zero for j >= n_actions) as the B element.  The contraction order inside a 16-wide slice is therefore permuted (slot g of MFMA
// (s, c) is column 16 s + 4 g + c) - a sum over k does not care, both operands agree.  H / 16 loads per lane are issued before the
// first MFMA: 16 KB per wavefront in flight.  The weights live in registers (H / 4 per lane).
#include "common.h"

namespace uavgnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NS>   // NS = H / 16 float4 loads per lane and row tile
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ h, int ld_h, int N, const float* __restrict__ W, int ld_w,
                                                       const float* __restrict__ b, int A, float* __restrict__ q, int ld_q, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  float4 w[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s)
    w[s] = j < A ? *reinterpret_cast<const float4*>(W + static_cast<size_t>(j) * ld_w + 16 * s + 4 * g) : float4{0.f, 0.f, 0.f, 0.f};
  const float bj = j < A ? b[j] : 0.f;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int row0 = tile * 16;
    const float* __restrict__ hr = h + static_cast<size_t>(min(row0 + j, N - 1)) * ld_h + 4 * g;
    float4 a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const float4*>(hr + 16 * s);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, w[s].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, w[s].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, w[s].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, w[s].w, acc, 0, 0, 0);
    }
    // D layout: lane (j, g) holds column j of rows 4 g .. 4 g + 3
    if (j < A) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row < N) q[static_cast<size_t>(row) * ld_q + j] = acc[r] + bj;
      }
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_head_supported(int H, int A) { return (H == 64 || H == 128 || H == 256) && A >= 1 && A <= 16; }

// q [N, A] (row stride ld_q) = h [N, H] (row stride ld_h) W [A, H]^T (row stride ld_w) + b [A]; fp32, products exact in fp32
extern "C" int uavgnn_head_fwd(const float* h, int ld_h, int N, int H, const float* W, int ld_w, const float* b, int A, float* q, int ld_q,
                               uavgnn_stream_t stream) {
  if (N < 0 || !h || !W || !b || !q || ld_h < H || ld_w < H || ld_q < A) return UAVGNN_EINVAL;
  if (!uavgnn_head_supported(H, A) || (ld_h & 3) || (ld_w & 3) ||
      ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(W)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const int tiles = (N + 15) / 16;
  const dim3 grid(capped_grid(tiles, 4, 4096)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define UAVGNN_HEAD(NS_) hipLaunchKernelGGL((head_fwd_kernel<NS_>), grid, block, 0, st, h, ld_h, N, W, ld_w, b, A, q, ld_q, tiles)
  switch (H) {
    case 64: UAVGNN_HEAD(4); break;
    case 128: UAVGNN_HEAD(8); break;
    default: UAVGNN_HEAD(16); break;
  }
#undef UAVGNN_HEAD
  return launch_status();
}
