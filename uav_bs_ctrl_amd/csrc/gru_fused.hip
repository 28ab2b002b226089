// K4 as SURVEY 2.2 specifies it: the WHOLE GRU cell in one kernel - both GEMMs on fp32 MFMA accumulating into r / z / n gate
// tiles, sigmoid / tanh / blend epilogue - so that the [N, 3H] pre-activation blocks gi, gh never reach HBM.
// Replaces nn.GRUCell at /root/reference/algos/madrqn/agents/gnn_agents.py:29,:123,:164,:208,:246,:282 (PyTorch gate order
// r, z, n):
//   r = sigma(W_ir i + b_ir + W_hr h + b_hr)   z = sigma(W_iz i + b_iz + W_hz h + b_hz)
//   n = tanh(W_in i + b_in + r (W_hn h + b_hn))   h' = (1 - z) n + z h
//
// One workgroup owns 128 rows (agents) x 32 hidden units.  Its four wavefronts (2 x 2) each hold 64 rows x 16 hidden units
// of FOUR accumulator sets - r and z (shared by both GEMMs: the sum is all the gates need), gi_n and gh_n - i.e. 4 x 4
// v_mfma_f32_16x16x4_f32 tiles = 64 accumulator registers, the shape of a plain 64 x 64 GEMM wave tile.  Phase 1 walks K
// over the input [x || c] with the r / z / n rows of W_ih as B operand, phase 2 walks K over h with the rows of W_hh; per
// 32-wide K slice the operands are staged in LDS (rows padded to 34 floats: conflict-free dword fragment reads), the next
// slice is in flight in registers while the current one computes.  The epilogue applies biases and gate math in registers
// on the MFMA D layout and stores h'; the training variant also stores the four pre-activation sets [N, 4H]
// (r_pre | z_pre | gi_n | gh_n, biases included) for uavgnn_gru_gates_bwd_fused.
//
// GEMM core measured stand-alone (tools/ubench/gemm_f32.hip, profiles/r02_ubench_gemm_f32_vs_rocblas.txt): 104-107 TFLOP/s
// at the GRU shapes = 0.98-1.00 of rocBLAS, 0.92-0.95 of the recorded hipBLASLt solutions; the fused cell wins by what it
// does NOT do: the 44 us gate pass and 8 KB per agent of gi / gh traffic.
#include "common.h"

namespace uavgnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BJ = 32, BK = 32, ST = 34;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void gru_cell_fwd_kernel(
    const float* __restrict__ inp, int ld_inp, int K1, const float* __restrict__ h, int N, int H,
    const float* __restrict__ W_ih, const float* __restrict__ b_ih, const float* __restrict__ W_hh,
    const float* __restrict__ b_hh, float* __restrict__ h_out, float* __restrict__ pre) {
  __shared__ __attribute__((aligned(16))) float sA[BM * ST];
  __shared__ __attribute__((aligned(16))) float sB[3 * BJ * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = (wave >> 1) * 64, wc = (wave & 1) * 16;
  const int m0 = blockIdx.y * BM, j0 = blockIdx.x * BJ;

  f32x4 acc[4][4];     // [row tile][set: r, z, gi_n, gh_n]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // loaders: A tile 128 rows x 32 k (thread -> row tid/2, 16 k), B tile 96 rows x 32 k (three float4 per thread)
  const int lr = tid >> 1, lk = (tid & 1) * 16;
  const int arow = min(m0 + lr, N - 1);                  // rows past N: clamped loads, masked stores
  float4 ra[4], rb[3];
  auto gload = [&](const float* __restrict__ Asrc, int lda, const float* __restrict__ Wsrc, int K, int k0) {
    const float* pa = Asrc + static_cast<size_t>(arow) * lda + k0 + lk;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = reinterpret_cast<const float4*>(pa)[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;   // row = set * 32 + unit
      const int wrow = (row >> 5) * H + j0 + (row & 31);
      rb[q] = *reinterpret_cast<const float4*>(Wsrc + static_cast<size_t>(wrow) * K + k0 + 4 * c4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* p = sA + lr * ST + lk + 4 * q;
      *reinterpret_cast<float2*>(p) = make_float2(ra[q].x, ra[q].y);
      *reinterpret_cast<float2*>(p + 2) = make_float2(ra[q].z, ra[q].w);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;
      float* p = sB + row * ST + 4 * c4;
      *reinterpret_cast<float2*>(p) = make_float2(rb[q].x, rb[q].y);
      *reinterpret_cast<float2*>(p + 2) = make_float2(rb[q].z, rb[q].w);
    }
  };

  // ---- phase 1: input GEMM (sets r, z, gi_n) -------------------------------------------------------------------------
  gload(inp, ld_inp, W_ih, K1, 0);
  for (int k0 = 0; k0 < K1; k0 += BK) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (k0 + BK < K1) gload(inp, ld_inp, W_ih, K1, k0 + BK);
    else gload(h, H, W_hh, H, 0);                         // first slice of phase 2
    // fragment reads: lane group g owns k = 8 p + 2 g + {0, 1} of every 8-wide sub-slice p (the contraction order is free as
    // long as both operands use the same mapping), so ONE ds_read_b64 serves two k-steps
#pragma unroll
    for (int ks = 0; ks < BK; ks += 8) {
      float2 fa[4], fb[3];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const float2*>(sA + (wm + a * 16 + j) * ST + ks + 2 * g);
#pragma unroll
      for (int s = 0; s < 3; ++s) fb[s] = *reinterpret_cast<const float2*>(sB + (s * BJ + wc + j) * ST + ks + 2 * g);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          acc[a][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].x, fb[s].x, acc[a][s], 0, 0, 0);
          acc[a][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].y, fb[s].y, acc[a][s], 0, 0, 0);
        }
    }
  }
  // ---- phase 2: hidden GEMM (sets r, z, gh_n) ------------------------------------------------------------------------
  for (int k0 = 0; k0 < H; k0 += BK) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (k0 + BK < H) gload(h, H, W_hh, H, k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 8) {
      float2 fa[4], fb[3];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const float2*>(sA + (wm + a * 16 + j) * ST + ks + 2 * g);
#pragma unroll
      for (int s = 0; s < 3; ++s) fb[s] = *reinterpret_cast<const float2*>(sB + (s * BJ + wc + j) * ST + ks + 2 * g);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].x, fb[0].x, acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].x, fb[1].x, acc[a][1], 0, 0, 0);
        acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].x, fb[2].x, acc[a][3], 0, 0, 0);
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].y, fb[0].y, acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].y, fb[1].y, acc[a][1], 0, 0, 0);
        acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].y, fb[2].y, acc[a][3], 0, 0, 0);
      }
    }
  }
  // ---- epilogue on the D layout: lane (g, j) holds rows 4g..4g+3 of every row tile, hidden unit c ------------------------
  // The h tile [128 x 32] comes in and the h' tile goes out through LDS with 16-byte row-contiguous accesses (the D layout
  // alone would touch HBM in 64-byte pieces).
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;
    const float4 t = *reinterpret_cast<const float4*>(h + static_cast<size_t>(min(m0 + row, N - 1)) * H + j0 + 4 * c4);
    float* p = sA + row * ST + 4 * c4;
    *reinterpret_cast<float2*>(p) = make_float2(t.x, t.y);
    *reinterpret_cast<float2*>(p + 2) = make_float2(t.z, t.w);
  }
  __syncthreads();
  const int c = j0 + wc + j;
  const float b_r = b_ih[c] + b_hh[c], b_z = b_ih[H + c] + b_hh[H + c], b_in = b_ih[2 * H + c], b_hn = b_hh[2 * H + c];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = wm + a * 16 + 4 * g + r;
      const int row = m0 + lrow;
      const float pr = acc[a][0][r] + b_r, pz = acc[a][1][r] + b_z, gin = acc[a][2][r] + b_in, ghn = acc[a][3][r] + b_hn;
      const float rr = sigmoidf_(pr), zz = sigmoidf_(pz);
      const float nn = tanhf(fmaf(rr, ghn, gin));
      float* hp = sA + lrow * ST + wc + j;
      *hp = fmaf(zz, *hp - nn, nn);                      // every element of the tile has exactly one owner lane
      if (SAVE && row < N) {
        float* p = pre + static_cast<size_t>(row) * 4 * H + c;
        p[0] = pr;
        p[H] = pz;
        p[2 * H] = gin;
        p[3 * H] = ghn;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;
    if (m0 + row < N) {
      const float* p = sA + row * ST + 4 * c4;
      *reinterpret_cast<float4*>(h_out + static_cast<size_t>(m0 + row) * H + j0 + 4 * c4) = make_float4(p[0], p[1], p[2], p[3]);
    }
  }
}

// pointwise backward from the saved pre-activation sets: d_gi [N,3H], d_gh [N,3H], d_h [N,H] (layout of gru.hip's kernel)
// HEAD: the gradient of h' is d_hout (or 0) + dq W_out, the head's input gradient formed on the fly (n_out <= 16 FMAs per element
// against rows of W_out that stay in L1): the [N, H] sum never goes through HBM and the rank-n_out GEMM launch disappears.
// col_sums (optional, H / 4 divides 256): [gridDim.x][4 H] column sums of this block's elements - d_r | d_z | d_n (input side) | d_n
// (hidden side), i.e. the block's share of the bias gradients db_ih = [0 : 3H], db_hh = [0 : 2H] + [3H : 4H] - so that the [T N, 3H]
// gate gradients of a BPTT sequence are not read again just to be summed (two streaming passes over 6.8 GB per C3 update).  A thread
// keeps its four columns over the grid-stride loop (the stride is a multiple of H / 4 elements), the 256 / (H / 4) threads of a
// block that share them are added in a fixed order through LDS: deterministic.
// RM (H = 256 only: the 64 lanes of a wavefront own ONE row): row_absmax[row] = max |.| over the row of d_gi and of d_gh - the row
// scales of the f16x2 products d x = [d_gi || d_proj] W and d h += d_gh W_hh behind this kernel (csrc/gemm_h2.hip).
template <bool HEAD, bool SUMS, bool RM = false>
__global__ __launch_bounds__(256) void gru_gates_bwd_fused_kernel(const float* __restrict__ pre, const float* __restrict__ h,
                                                                  const float* __restrict__ d_hout, long long total, int H,
                                                                  float* __restrict__ d_gi, float* __restrict__ d_gh,
                                                                  float* __restrict__ d_h, const float* __restrict__ dq,
                                                                  int n_out, const float* __restrict__ W_out,
                                                                  float* __restrict__ col_sums,
                                                                  float* __restrict__ row_absmax = nullptr) {
  const int HV = H / 4;
  float cs[SUMS ? 16 : 1];       // the accumulators (and the LDS array below) exist in the column-sum instantiations only
#pragma unroll
  for (int t = 0; t < (SUMS ? 16 : 1); ++t) cs[t] = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const long long row = i / HV;
    const int col = static_cast<int>(i - row * HV) * 4;
    const float* p = pre + row * 4 * H + col;
    const float4 pr = *reinterpret_cast<const float4*>(p), pz = *reinterpret_cast<const float4*>(p + H);
    const float4 gin = *reinterpret_cast<const float4*>(p + 2 * H), ghn = *reinterpret_cast<const float4*>(p + 3 * H);
    const float4 hh = *reinterpret_cast<const float4*>(h + row * H + col);
    float4 dho = d_hout != nullptr ? *reinterpret_cast<const float4*>(d_hout + row * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (HEAD) {
      const float* __restrict__ qr = dq + row * n_out;
      for (int a = 0; a < n_out; ++a) {
        const float qa = qr[a];
        const float4 w = *reinterpret_cast<const float4*>(W_out + static_cast<size_t>(a) * H + col);
        dho.x = fmaf(qa, w.x, dho.x);
        dho.y = fmaf(qa, w.y, dho.y);
        dho.z = fmaf(qa, w.z, dho.z);
        dho.w = fmaf(qa, w.w, dho.w);
      }
    }
    const float a_pr[4] = {pr.x, pr.y, pr.z, pr.w}, a_pz[4] = {pz.x, pz.y, pz.z, pz.w};
    const float a_gi[4] = {gin.x, gin.y, gin.z, gin.w}, a_gh[4] = {ghn.x, ghn.y, ghn.z, ghn.w};
    const float a_h[4] = {hh.x, hh.y, hh.z, hh.w}, a_d[4] = {dho.x, dho.y, dho.z, dho.w};
    float dr[4], dz[4], dni[4], dnh[4], dh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float r = sigmoidf_(a_pr[t]), z = sigmoidf_(a_pz[t]);
      const float n = tanhf(fmaf(r, a_gh[t], a_gi[t]));
      const float dn_pre = a_d[t] * (1.f - z) * (1.f - n * n);
      dni[t] = dn_pre;
      dnh[t] = dn_pre * r;
      dr[t] = dn_pre * a_gh[t] * r * (1.f - r);
      dz[t] = a_d[t] * (a_h[t] - n) * z * (1.f - z);
      dh[t] = a_d[t] * z;
      if constexpr (SUMS) {
        cs[t] += dr[t];
        cs[4 + t] += dz[t];
        cs[8 + t] += dni[t];
        cs[12 + t] += dnh[t];
      }
    }
    float* gi = d_gi + row * 3 * H + col;
    float* gh = d_gh + row * 3 * H + col;
    *reinterpret_cast<float4*>(gi) = make_float4(dr[0], dr[1], dr[2], dr[3]);
    *reinterpret_cast<float4*>(gi + H) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    *reinterpret_cast<float4*>(gi + 2 * H) = make_float4(dni[0], dni[1], dni[2], dni[3]);
    *reinterpret_cast<float4*>(gh) = make_float4(dr[0], dr[1], dr[2], dr[3]);
    *reinterpret_cast<float4*>(gh + H) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    *reinterpret_cast<float4*>(gh + 2 * H) = make_float4(dnh[0], dnh[1], dnh[2], dnh[3]);
    *reinterpret_cast<float4*>(d_h + row * H + col) = make_float4(dh[0], dh[1], dh[2], dh[3]);
    if constexpr (RM) {
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) m = fmaxf(fmaxf(fmaxf(m, fabsf(dr[t])), fmaxf(fabsf(dz[t]), fabsf(dni[t]))), fabsf(dnh[t]));
      m = wave_max(m);
      if ((threadIdx.x & 63) == 0) row_absmax[row] = m;
    }
  }
  if constexpr (SUMS) {
    __shared__ float sS[256 * 17];
#pragma unroll
    for (int t = 0; t < 16; ++t) sS[threadIdx.x * 17 + t] = cs[t];
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < HV) {
      float* out = col_sums + static_cast<size_t>(blockIdx.x) * 4 * H + 4 * threadIdx.x;
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) {
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float a = 0.f;
          for (int part = threadIdx.x; part < 256; part += HV) a += sS[part * 17 + 4 * gate + t];
          v[t] = a;
        }
        *reinterpret_cast<float4*>(out + gate * H) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_gru_cell_supported(int K_in, int H) {
  return (K_in >= BK && K_in % BK == 0 && H >= BK && H % BK == 0) ? 1 : 0;
}

extern "C" int uavgnn_gru_cell_fwd(const float* inp, int ld_inp, int K_in, const float* h, int N, int H, const float* W_ih,
                                   const float* b_ih, const float* W_hh, const float* b_hh, float* h_out, float* pre_save,
                                   uavgnn_stream_t stream) {
  if (N < 0 || !inp || !h || !W_ih || !b_ih || !W_hh || !b_hh || !h_out || ld_inp < K_in) return UAVGNN_EINVAL;
  if (!uavgnn_gru_cell_supported(K_in, H) || (ld_inp & 3) ||
      ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(W_ih) |
        reinterpret_cast<uintptr_t>(W_hh)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const dim3 grid(H / BJ, (N + BM - 1) / BM), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pre_save != nullptr)
    hipLaunchKernelGGL(gru_cell_fwd_kernel<true>, grid, block, 0, st, inp, ld_inp, K_in, h, N, H, W_ih, b_ih, W_hh, b_hh, h_out,
                       pre_save);
  else
    hipLaunchKernelGGL(gru_cell_fwd_kernel<false>, grid, block, 0, st, inp, ld_inp, K_in, h, N, H, W_ih, b_ih, W_hh, b_hh, h_out,
                       pre_save);
  return launch_status();
}

extern "C" int uavgnn_gru_gates_bwd_fused(const float* pre, const float* h, const float* d_hout, int N, int H, float* d_gi,
                                          float* d_gh, float* d_h, uavgnn_stream_t stream) {
  if (N < 0 || H <= 0 || !pre || !h || !d_hout || !d_gi || !d_gh || !d_h) return UAVGNN_EINVAL;
  if (H % 4) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const long long total = static_cast<long long>(N) * (H / 4);
  hipLaunchKernelGGL((gru_gates_bwd_fused_kernel<false, false>), dim3(capped_grid(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pre, h, d_hout, total, H, d_gi, d_gh, d_h, nullptr, 0, nullptr, nullptr);
  return launch_status();
}

extern "C" int uavgnn_gru_gates_bwd_fused_head(const float* pre, const float* h, const float* d_hout, const float* dq, int n_out,
                                               const float* W_out, int N, int H, float* d_gi, float* d_gh, float* d_h,
                                               uavgnn_stream_t stream) {
  if (N < 0 || H <= 0 || n_out <= 0 || !pre || !h || !dq || !W_out || !d_gi || !d_gh || !d_h) return UAVGNN_EINVAL;
  if (H % 4 || n_out > 64 || (reinterpret_cast<uintptr_t>(W_out) & 15)) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const long long total = static_cast<long long>(N) * (H / 4);
  hipLaunchKernelGGL((gru_gates_bwd_fused_kernel<true, false>), dim3(capped_grid(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pre, h, d_hout, total, H, d_gi, d_gh, d_h, dq, n_out, W_out, nullptr);
  return launch_status();
}

// rows of the `col_sums` matrix uavgnn_gru_gates_bwd_fused_sums writes for N rows (0: this H has no column-sum variant)
extern "C" int uavgnn_gru_gates_bwd_sum_rows(int N, int H) {
  if (N <= 0 || H <= 0 || (H % 4) || (256 % (H / 4))) return 0;
  return capped_grid(static_cast<long long>(N) * (H / 4), 256);
}

// ... the same launch (dq / W_out may be NULL: no head term) that ALSO writes col_sums [uavgnn_gru_gates_bwd_sum_rows(N, H)][4 H]:
// per-block column sums d_r | d_z | d_n (input side) | d_n (hidden side) - the bias gradients of the cell are their sum over rows
extern "C" int uavgnn_gru_gates_bwd_fused_sums(const float* pre, const float* h, const float* d_hout, const float* dq, int n_out,
                                               const float* W_out, int N, int H, float* d_gi, float* d_gh, float* d_h,
                                               float* col_sums, uavgnn_stream_t stream) {
  if (N < 0 || H <= 0 || !pre || !h || !d_gi || !d_gh || !d_h || !col_sums || (dq == nullptr) != (W_out == nullptr) ||
      (dq == nullptr && d_hout == nullptr) || (dq != nullptr && n_out <= 0))
    return UAVGNN_EINVAL;
  if (N == 0) return 0;
  if (!uavgnn_gru_gates_bwd_sum_rows(N, H) || n_out > 64 || ((reinterpret_cast<uintptr_t>(W_out) | reinterpret_cast<uintptr_t>(col_sums)) & 15))
    return UAVGNN_EUNSUPPORTED;
  const long long total = static_cast<long long>(N) * (H / 4);
  const dim3 grid(capped_grid(total, 256)), block(256);
  if (dq != nullptr)
    hipLaunchKernelGGL((gru_gates_bwd_fused_kernel<true, true>), grid, block, 0, static_cast<hipStream_t>(stream), pre, h, d_hout, total, H,
                       d_gi, d_gh, d_h, dq, n_out, W_out, col_sums);
  else
    hipLaunchKernelGGL((gru_gates_bwd_fused_kernel<false, true>), grid, block, 0, static_cast<hipStream_t>(stream), pre, h, d_hout, total, H,
                       d_gi, d_gh, d_h, nullptr, 0, nullptr, col_sums);
  return launch_status();
}

// ... that ALSO writes row_absmax [N] = max |.| over the rows of d_gi and d_gh (H = 256: one wavefront per row; UAVGNN_EUNSUPPORTED
// otherwise): the row scales of the f16x2 input-gradient products behind it (uavgnn_gemm_nt_h2)
extern "C" int uavgnn_gru_gates_bwd_fused_sums_rowmax(const float* pre, const float* h, const float* d_hout, const float* dq, int n_out,
                                                      const float* W_out, int N, int H, float* d_gi, float* d_gh, float* d_h,
                                                      float* col_sums, float* row_absmax, uavgnn_stream_t stream) {
  if (N < 0 || H <= 0 || !pre || !h || !d_gi || !d_gh || !d_h || !col_sums || !row_absmax || (dq == nullptr) != (W_out == nullptr) ||
      (dq == nullptr && d_hout == nullptr) || (dq != nullptr && n_out <= 0))
    return UAVGNN_EINVAL;
  if (N == 0) return 0;
  if (H != 256 || !uavgnn_gru_gates_bwd_sum_rows(N, H) || n_out > 64 ||
      ((reinterpret_cast<uintptr_t>(W_out) | reinterpret_cast<uintptr_t>(col_sums)) & 15))
    return UAVGNN_EUNSUPPORTED;
  const long long total = static_cast<long long>(N) * (H / 4);
  const dim3 grid(capped_grid(total, 256)), block(256);
  if (dq != nullptr)
    hipLaunchKernelGGL((gru_gates_bwd_fused_kernel<true, true, true>), grid, block, 0, static_cast<hipStream_t>(stream), pre, h, d_hout, total,
                       H, d_gi, d_gh, d_h, dq, n_out, W_out, col_sums, row_absmax);
  else
    hipLaunchKernelGGL((gru_gates_bwd_fused_kernel<false, true, true>), grid, block, 0, static_cast<hipStream_t>(stream), pre, h, d_hout,
                       total, H, d_gi, d_gh, d_h, nullptr, 0, nullptr, col_sums, row_absmax);
  return launch_status();
}
