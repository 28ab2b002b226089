// K4 on the bf16 matrix cores: the whole GRU cell in one kernel (as gru_fused.hip), with the two GEMMs computed as exact
// three-way bf16 splits of the fp32 operands and six bf16 MFMA products per fp32 product (bf16x3.h: fp32-level error, below
// rocBLAS sgemm's on the same data; fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950).
// Replaces nn.GRUCell at /root/reference/algos/madrqn/agents/gnn_agents.py:29,:123,:164,:208,:246,:282 (gate order r, z, n):
//   r = sigma(W_ir i + b_ir + W_hr h + b_hr)   z = sigma(W_iz i + b_iz + W_hz h + b_hz)
//   n = tanh(W_in i + b_in + r (W_hn h + b_hn))   h' = (1 - z) n + z h
//
// Data flow.  uavgnn_gru_split_weights turns W_ih [3H, K_in] and W_hh [3H, H] into bf16 planes [3][3H][K] (one launch per
// forward: the weights of a drop-in module may change behind any cache - `.data` writes do not bump a version counter - and
// the launch costs 3 us).  gru_cell_fwd_x3_kernel: one workgroup owns 128 agents x 32 hidden units and four accumulator
// sets r, z, gi_n, gh_n (r and z take both contractions); per 32-wide K slice the activation tile is loaded as fp32, split
// in registers (v_cvt_pk_bf16_f32 + two subtractions per term) and written to LDS as three bf16 planes, the weight planes
// are copied; LDS rows are 64 B with an XOR swizzle (conflict-free ds_read_b128 fragments, no padding); every wavefront
// (64 agents x 16 units) issues 72 v_mfma_f32_16x16x32_bf16 per slice.  The eight column blocks of a row block run on the
// same XCD (blockIdx -> XCD round robin), so the activation tile is fetched from HBM once and re-read from that XCD's L2.
// Epilogue as gru_fused.hip: biases, gates, h' through an LDS tile (full-row HBM accesses), optional pre-activations
// [N, 4H] for uavgnn_gru_gates_bwd_fused.
#include "bf16x3.h"
#include "common.h"

namespace uavgnn {
namespace {

using namespace x3;
constexpr int BM = 128, BJ = 32, BK = 32, ST = 34;
constexpr int PA = BM * 4, PB = 3 * BJ * 4;
#ifndef SKEW
#define SKEW 32   // x 64 cycles
#endif   // 16-byte chunks per split plane of the A / B tile

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// two weight matrices -> bf16 planes, one launch: pair index p over (n0 + n1) / 2
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ w0, unsigned short* __restrict__ p0,
                                                           long long n0, const float* __restrict__ w1,
                                                           unsigned short* __restrict__ p1, long long n1) {
  long long i = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 2;
  const float* w = w0;
  unsigned short* p = p0;
  long long n = n0;
  if (i >= n0) {
    i -= n0;
    w = w1;
    p = p1;
    n = n1;
    if (i >= n1) return;
  }
  const float2 v = *reinterpret_cast<const float2*>(w + i);
  const Split3 s = split_pair(v.x, v.y);
  *reinterpret_cast<unsigned*>(p + i) = s.h1;
  *reinterpret_cast<unsigned*>(p + n + i) = s.h2;
  *reinterpret_cast<unsigned*>(p + 2 * n + i) = s.h3;
}

// NK1 / NK2 > 0: the slice counts K_in / 32 and H / 32 are compile-time, the slice loop is fully unrolled and the loads run
// TWO slices ahead (with a rolled loop the compiler's s_waitcnt placement drains every load at the loop header, i.e. a
// prefetch distance of one compute phase - shorter than the loaded L2 / HBM latency: 45 % of the wave cycles were s_waitcnt).
template <bool SAVE, int NK1, int NK2>
__global__ __launch_bounds__(256, 2) void gru_cell_fwd_x3_kernel(
    const float* __restrict__ inp, int ld_inp, int K1, const float* __restrict__ h, int N, int H,
    const unsigned short* __restrict__ Wih_p, const float* __restrict__ b_ih, const unsigned short* __restrict__ Whh_p,
    const float* __restrict__ b_hh, float* __restrict__ h_out, float* __restrict__ pre, int row_blocks) {
  __shared__ u32x4 sA[3 * PA];   // [plane][128 rows][4 chunks]          24 KB (reused as the fp32 h / h' tile)
  __shared__ u32x4 sB[3 * PB];   // [plane][96 rows = gate * 32 + unit][4] 18 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = (wave >> 1) * 64, wc = (wave & 1) * 16;
  const int CB = H / BJ;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / CB) * 8 + xcd, cb = slot - (slot / CB) * CB;
  if (rb >= row_blocks) return;
  const int m0 = rb * BM, j0 = cb * BJ;
  // Two workgroups share a CU and alternate a VALU phase (split + LDS writes) with an MFMA phase per slice; started together
  // they stay in lockstep and the matrix pipe idles through both VALU phases.  The second workgroup of every CU in the first
  // round starts half a slice late, later rounds inherit the skew (a CU's slots now free up at different times).
  if (blockIdx.x >= 256 && blockIdx.x < 512) __builtin_amdgcn_s_sleep(SKEW);

  f32x4 acc[4][4];     // [row tile][set: r, z, gi_n, gh_n]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A loader: float4 q = tid + 256 i -> row tid / 8 + 32 i, k = 4 (tid % 8); rows past N are clamped (stores are masked)
  const int lr = tid >> 3, c4 = tid & 7;
  int ar[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ar[i] = min(m0 + lr + 32 * i, N - 1);
  unsigned short* sa_w = reinterpret_cast<unsigned short*>(sA) + lr * 32 + (((c4 >> 1) ^ swz(lr)) * 8) + (c4 & 1) * 4;
  // B loader: chunk q = tid + 256 i (i < 5, 1152 chunks): plane q / 384, row (q % 384) / 4 = gate * 32 + unit, chunk q % 4
  int wr[5], wk[5], sbw[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int q = min(tid + 256 * i, 3 * PB - 1), pl = q / PB, rem = q - pl * PB, row = rem >> 2, c = rem & 3;
    wr[i] = pl * 3 * H + (row >> 5) * H + j0 + (row & 31);
    wk[i] = 8 * c;
    sbw[i] = pl * PB + row * 4 + (c ^ swz(row));
  }
  float4 ra[2][4];
  u32x4 rw[2][5];
#define UAVGNN_X3_GLOAD(SET, Asrc, lda, Wp, K, k0)                                                                  \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[SET][i] =                                                     \
        *reinterpret_cast<const float4*>((Asrc) + static_cast<size_t>(ar[i]) * (lda) + (k0) + 4 * c4);              \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) rw[SET][i] =                                                     \
        *reinterpret_cast<const u32x4*>((Wp) + static_cast<size_t>(wr[i]) * (K) + wk[i] + (k0));                    \
  }
#if defined(UAVGNN_X3_DBG) && UAVGNN_X3_DBG == 3   /* timing experiment: raw copies, no split */
#define UAVGNN_X3_LSTORE(SET)                                                                                      \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) sA[tid + 256 * i] = __builtin_bit_cast(u32x4, ra[SET][i]);        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) sB[sbw[i]] = rw[SET][i];                                         \
    if (tid < 3 * PB - 1024) sB[sbw[4]] = rw[SET][4];                                                              \
  }
#elif defined(UAVGNN_X3_DBG) && UAVGNN_X3_DBG == 4   /* timing experiment: nothing staged at all */
#define UAVGNN_X3_LSTORE(SET) {}
#else
#define UAVGNN_X3_LSTORE(SET)                                                                                      \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) stage4(sa_w + 32 * i * 32, PA * 8, ra[SET][i]);                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) sB[sbw[i]] = rw[SET][i];                                         \
    if (tid < 3 * PB - 1024) sB[sbw[4]] = rw[SET][4];                                                              \
  }
#endif
  // one K slice: 4 row tiles x 3 gates x 6 products; smallest products first, three independent accumulators per product
#define UAVGNN_X3_TERM(ia, ib, NS)                \
  acc[a][0] = mfma(fa[ia], fb[0][ib], acc[a][0]); \
  acc[a][1] = mfma(fa[ia], fb[1][ib], acc[a][1]); \
  acc[a][NS] = mfma(fa[ia], fb[2][ib], acc[a][NS]);
#if defined(UAVGNN_X3_DBG) && UAVGNN_X3_DBG == 1
#define UAVGNN_X3_SLICE(NS) {}
#else
#define UAVGNN_X3_SLICE(NS)                                                                                        \
  {                                                                                                                \
    bf16x8 fb[3][3];                                                                                               \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) fb[s][pl] =     \
        as_frag(sB[pl * PB + (s * BJ + wc + j) * 4 + (g ^ swz(j))]);                                               \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                                \
      bf16x8 fa[3];                                                                                                \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) fa[pl] = as_frag(sA[pl * PA + (wm + a * 16 + j) * 4 + (g ^ swz(j))]); \
      UAVGNN_X3_TERM(0, 2, NS) UAVGNN_X3_TERM(2, 0, NS) UAVGNN_X3_TERM(1, 1, NS)                                  \
      UAVGNN_X3_TERM(0, 1, NS) UAVGNN_X3_TERM(1, 0, NS) UAVGNN_X3_TERM(0, 0, NS)                                  \
    }                                                                                                              \
  }
#endif

  if constexpr (NK1 > 0) {
    // slice t < NK1: input GEMM (sets r, z, gi_n); t >= NK1: hidden GEMM (sets r, z, gh_n); loads two slices ahead
    constexpr int NS = NK1 + NK2;
#if defined(UAVGNN_X3_DBG) && UAVGNN_X3_DBG == 2   /* timing experiment: only the first two slices are ever loaded */
#define UAVGNN_X3_LOAD_SLICE(SET, t) \
  if ((t) < 2) UAVGNN_X3_GLOAD(SET, inp, ld_inp, Wih_p, K1, (t) * BK)
#else
#define UAVGNN_X3_LOAD_SLICE(SET, t)                                          \
  if ((t) < NK1) UAVGNN_X3_GLOAD(SET, inp, ld_inp, Wih_p, K1, (t) * BK)       \
  else UAVGNN_X3_GLOAD(SET, h, H, Whh_p, H, ((t) - NK1) * BK)
#endif
    UAVGNN_X3_LOAD_SLICE(0, 0)
    UAVGNN_X3_LOAD_SLICE(1, 1)
#pragma unroll
    for (int t = 0; t < NS; t += 2) {
      __syncthreads();
      UAVGNN_X3_LSTORE(0)
      __syncthreads();
      if (t + 2 < NS) UAVGNN_X3_LOAD_SLICE(0, t + 2)
      if (t < NK1) UAVGNN_X3_SLICE(2) else UAVGNN_X3_SLICE(3)
      if (t + 1 < NS) {
        __syncthreads();
        UAVGNN_X3_LSTORE(1)
        __syncthreads();
        if (t + 3 < NS) UAVGNN_X3_LOAD_SLICE(1, t + 3)
        if (t + 1 < NK1) UAVGNN_X3_SLICE(2) else UAVGNN_X3_SLICE(3)
      }
    }
#undef UAVGNN_X3_LOAD_SLICE
  } else {
    // ---- phase 1: input GEMM (sets r, z, gi_n) -----------------------------------------------------------------------
    UAVGNN_X3_GLOAD(0, inp, ld_inp, Wih_p, K1, 0)
    for (int k0 = 0; k0 < K1; k0 += BK) {
      __syncthreads();
      UAVGNN_X3_LSTORE(0)
      __syncthreads();
      if (k0 + BK < K1) UAVGNN_X3_GLOAD(0, inp, ld_inp, Wih_p, K1, k0 + BK)
      else UAVGNN_X3_GLOAD(0, h, H, Whh_p, H, 0)            // first slice of phase 2
      UAVGNN_X3_SLICE(2)
    }
    // ---- phase 2: hidden GEMM (sets r, z, gh_n) ----------------------------------------------------------------------
    for (int k0 = 0; k0 < H; k0 += BK) {
      __syncthreads();
      UAVGNN_X3_LSTORE(0)
      __syncthreads();
      UAVGNN_X3_GLOAD(0, h, H, Whh_p, H, min(k0 + BK, H - BK))   // unconditional (the tail re-reads the last slice)
      UAVGNN_X3_SLICE(3)
    }
  }
#undef UAVGNN_X3_GLOAD
#undef UAVGNN_X3_LSTORE
#undef UAVGNN_X3_SLICE
#undef UAVGNN_X3_TERM
  // ---- epilogue on the D layout: lane (g, j) holds rows 4g..4g+3 of every row tile, hidden unit c ------------------------
  float* sH = reinterpret_cast<float*>(sA);               // [128][ST] fp32 tile: h in, h' out, 16-byte row-contiguous HBM accesses
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = lr + 32 * q;
    const float4 t = *reinterpret_cast<const float4*>(h + static_cast<size_t>(ar[q]) * H + j0 + 4 * c4);
    float* p = sH + row * ST + 4 * c4;
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
      float* hp = sH + lrow * ST + wc + j;
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
    const int row = lr + 32 * q;
    if (m0 + row < N) {
      const float* p = sH + row * ST + 4 * c4;
      *reinterpret_cast<float4*>(h_out + static_cast<size_t>(m0 + row) * H + j0 + 4 * c4) = make_float4(p[0], p[1], p[2], p[3]);
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" long long uavgnn_gru_cell_x3_workspace_bytes(int K_in, int H) {
  if (K_in <= 0 || H <= 0) return 0;
  return 3LL * 3 * H * (static_cast<long long>(K_in) + H) * 2;
}

extern "C" int uavgnn_gru_split_weights(const float* W_ih, int K_in, const float* W_hh, int H, void* planes,
                                        uavgnn_stream_t stream) {
  if (!W_ih || !W_hh || !planes || K_in <= 0 || H <= 0) return UAVGNN_EINVAL;
  if ((K_in & 1) || (H & 1) || ((reinterpret_cast<uintptr_t>(W_ih) | reinterpret_cast<uintptr_t>(W_hh)) & 7) ||
      (reinterpret_cast<uintptr_t>(planes) & 15))
    return UAVGNN_EUNSUPPORTED;
  const long long n0 = 3LL * H * K_in, n1 = 3LL * H * H;
  unsigned short* p0 = static_cast<unsigned short*>(planes);
  unsigned short* p1 = p0 + 3 * n0;
  const long long pairs = (n0 + n1) / 2;
  hipLaunchKernelGGL(split_planes_kernel, dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), W_ih, p0, n0, W_hh, p1, n1);
  return launch_status();
}

extern "C" int uavgnn_gru_cell_fwd_x3(const float* inp, int ld_inp, int K_in, const float* h, int N, int H,
                                      const void* planes, const float* b_ih, const float* b_hh, float* h_out,
                                      float* pre_save, uavgnn_stream_t stream) {
  if (N < 0 || !inp || !h || !planes || !b_ih || !b_hh || !h_out || ld_inp < K_in) return UAVGNN_EINVAL;
  if (!uavgnn_gru_cell_supported(K_in, H) || (ld_inp & 3) ||
      ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(planes) |
        reinterpret_cast<uintptr_t>(h_out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const unsigned short* p0 = static_cast<const unsigned short*>(planes);
  const unsigned short* p1 = p0 + 9LL * H * K_in;
  const int row_blocks = (N + BM - 1) / BM;
  const dim3 grid(((row_blocks + 7) / 8) * 8 * (H / BJ)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define UAVGNN_X3_LAUNCH(SAVE, NK1, NK2)                                                                                  \
  hipLaunchKernelGGL((gru_cell_fwd_x3_kernel<SAVE, NK1, NK2>), grid, block, 0, st, inp, ld_inp, K_in, h, N, H, p0, b_ih, p1, \
                     b_hh, h_out, pre_save, row_blocks)
  if (K_in == 320 && H == 256) {        // exp3 / C3 shape (msg 64 + obs 256 -> 256): unrolled, loads two slices ahead
    if (pre_save != nullptr) UAVGNN_X3_LAUNCH(true, 10, 8);
    else UAVGNN_X3_LAUNCH(false, 10, 8);
  } else {
    if (pre_save != nullptr) UAVGNN_X3_LAUNCH(true, 0, 0);
    else UAVGNN_X3_LAUNCH(false, 0, 0);
  }
#undef UAVGNN_X3_LAUNCH
  return launch_status();
}
