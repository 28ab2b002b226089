// Weight gradients of the dense layers on the bf16 matrix cores (bf16x3.h): P[s][Mo, Ko] (+)= dY[rows_s, :Mo]^T X[rows_s, :Ko],
// fp32 in and out, every fp32 product the fp32-accumulated sum of six exact bf16 x bf16 MFMA products (the arithmetic of
// gemm_x3.hip / gru_x3.hip).  Replaces autograd's dW = dy^T x of the nn.Linear / nn.GRUCell layers of
// /root/reference/algos/madrqn/agents/gnn_agents.py:99 (f_aggr), :243-246 (f_val / f_sign / f_que, f_udt), :43-46 (f_out) under
// loss.backward() (algos/madrqn/learner.py:157), which the round-2 build ran on the vendor's batched split-K fp32 GEMM.
//
// The contraction runs over the AGENT axis (10^4 .. 10^6 rows) while the output is at most 768 x 512, so
//   * the rows are cut into S chunks, one partial product P[s] per chunk (the caller sums the S partials in a fixed order:
//     deterministic, no atomics), `accumulate` adds into P[s] in place - the BPTT backward accumulates its T + 1 steps there;
//   * both operands are row-major with the contraction index as the SLOW index, the opposite of what an MFMA fragment wants
//     (8 consecutive k per lane).  The transposition happens on the way into LDS: a thread owns ONE output feature (column of
//     dY / X) and 16 consecutive rows of the 32-row slice - 16 dword loads, each of them a 256-byte coalesced segment per
//     wavefront - splits the values pairwise along k and writes two 16-byte k-chunks per bf16 plane.  The LDS image is the one
//     of gemm_x3.hip ([plane][feature][4 chunks of 8 k], XOR swizzle), so fragment reads and the MFMA block are the same.
// One workgroup: 128 x 128 output tile, four wavefronts of 64 x 64 on v_mfma_f32_32x32x16_bf16, 48 KB of LDS, the next slice in
// flight in registers while the current one computes.
#include "bf16x3.h"
#include "common.h"

namespace uavgnn {
namespace {

using namespace x3;
#ifndef GEMM_TN_XCD_ORDER
#define GEMM_TN_XCD_ORDER 0   // 1: all tiles of a row chunk on one XCD (measured with tools/gemm_tn_big_probe.py: +4 % on the 18-tile W_ih shape, -5 % on the 8-tile f_aggr shape, the one shape that runs here - the time-batched recurrent weights went to the vendor GEMM)
#endif
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PT = 128 * 4;   // 16-byte chunks per split plane of a 128-feature tile

template <bool ACC>
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(const float* __restrict__ Yd, int ldy, int Mo,
                                                            const float* __restrict__ X, int ldx, int Ko, long long n_rows,
                                                            long long chunk, float* __restrict__ P, int col_blocks) {
  __shared__ u32x4 sA[3 * PT], sB[3 * PT];   // [plane][feature][4 chunks of 8 bf16 along the row (contraction) index]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, lh = lane >> 5, sw = swz32(l32);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // Workgroup -> (output tile, row chunk).  Consecutive workgroup ids go to consecutive XCDs (8, each with its own L2), and
  // every tile of a row chunk reads the same rows of dY / X: all tiles of chunk s are therefore placed on XCD s % 8, next to
  // each other in launch order, so that a row slice is fetched from HBM once per chunk and not once per tile (time-batched
  // weight gradients: 18 tiles over 1.67 M rows moved 30 GB instead of 7 GB and ran at 120 instead of 160 TFLOP/s).
  // S is a multiple of 8 whenever it is at least 8 (uavgnn_gemm_tn_x3_chunks); smaller S keep the plain order.
  const int tiles = gridDim.x, S_all = gridDim.y;
  int tile = blockIdx.x, s = blockIdx.y;
  if (GEMM_TN_XCD_ORDER && (S_all & 7) == 0) {
    const int id = blockIdx.y * tiles + blockIdx.x;   // launch order: x fastest
    const int xcd = id & 7, k = id >> 3;
    tile = k % tiles;
    s = (k / tiles) * 8 + xcd;
  }
  const int rb = tile / col_blocks, cb = tile - rb * col_blocks;
  const int m0 = rb * BM, n0 = cb * BN;
  const long long r_begin = static_cast<long long>(s) * chunk, r_end = min(r_begin + chunk, n_rows);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // loader: thread -> feature f = tid % 128 of each tile, rows 16 kh .. 16 kh + 15 of the slice (kh = tid / 128); features past
  // Mo / Ko are clamped (their outputs are never stored), rows past the chunk are clamped and zeroed
  const int f = tid & 127, kh = tid >> 7;
  const float* pa = Yd + min(m0 + f, Mo - 1);
  const float* pb = X + min(n0 + f, Ko - 1);
  const int chunk_lo = ((2 * kh) ^ swz32(f)), chunk_hi = ((2 * kh + 1) ^ swz32(f));
  float ra[16], rb_[16];
  auto gload = [&](long long k0) {   // always 32 loads (clamped addresses)
    const long long base = k0 + 16 * kh;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long r = min(base + i, n_rows - 1);
      ra[i] = pa[r * ldy];
      rb_[i] = pb[r * ldx];
    }
  };
  auto lstore = [&](long long k0) {
    const long long base = k0 + 16 * kh;
    Split3 sa[8], sb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool v0 = base + 2 * i < r_end, v1 = base + 2 * i + 1 < r_end;
      sa[i] = split_pair(v0 ? ra[2 * i] : 0.f, v1 ? ra[2 * i + 1] : 0.f);
      sb[i] = split_pair(v0 ? rb_[2 * i] : 0.f, v1 ? rb_[2 * i + 1] : 0.f);
    }
    sA[0 * PT + f * 4 + chunk_lo] = u32x4{sa[0].h1, sa[1].h1, sa[2].h1, sa[3].h1};
    sA[0 * PT + f * 4 + chunk_hi] = u32x4{sa[4].h1, sa[5].h1, sa[6].h1, sa[7].h1};
    sA[1 * PT + f * 4 + chunk_lo] = u32x4{sa[0].h2, sa[1].h2, sa[2].h2, sa[3].h2};
    sA[1 * PT + f * 4 + chunk_hi] = u32x4{sa[4].h2, sa[5].h2, sa[6].h2, sa[7].h2};
    sA[2 * PT + f * 4 + chunk_lo] = u32x4{sa[0].h3, sa[1].h3, sa[2].h3, sa[3].h3};
    sA[2 * PT + f * 4 + chunk_hi] = u32x4{sa[4].h3, sa[5].h3, sa[6].h3, sa[7].h3};
    sB[0 * PT + f * 4 + chunk_lo] = u32x4{sb[0].h1, sb[1].h1, sb[2].h1, sb[3].h1};
    sB[0 * PT + f * 4 + chunk_hi] = u32x4{sb[4].h1, sb[5].h1, sb[6].h1, sb[7].h1};
    sB[1 * PT + f * 4 + chunk_lo] = u32x4{sb[0].h2, sb[1].h2, sb[2].h2, sb[3].h2};
    sB[1 * PT + f * 4 + chunk_hi] = u32x4{sb[4].h2, sb[5].h2, sb[6].h2, sb[7].h2};
    sB[2 * PT + f * 4 + chunk_lo] = u32x4{sb[0].h3, sb[1].h3, sb[2].h3, sb[3].h3};
    sB[2 * PT + f * 4 + chunk_hi] = u32x4{sb[4].h3, sb[5].h3, sb[6].h3, sb[7].h3};
  };

  if (r_begin < r_end) {
    gload(r_begin);
    for (long long k0 = r_begin; k0 < r_end; k0 += BK) {
      __syncthreads();
      lstore(k0);
      __syncthreads();
      gload(k0 + BK < r_end ? k0 + BK : k0);             // unconditional (the tail re-reads the last slice): static vmcnt
      __builtin_amdgcn_sched_barrier(0);                 // keep the loads ahead of the MFMA block
      bf16x8 fa[2][2][3], fb[2][2][3];                   // [tile][half][plane]
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            fa[a][h2][pl] = as_frag(sA[pl * PT + (wm + a * 32 + l32) * 4 + ((2 * h2 + lh) ^ sw)]);
            fb[a][h2][pl] = as_frag(sB[pl * PT + (wn + a * 32 + l32) * 4 + ((2 * h2 + lh) ^ sw)]);
          }
      // six products, smallest first; four independent accumulators between dependent MFMAs
#define UAVGNN_X3_TERM(ia, ib)                                                                     \
  _Pragma("unroll") for (int h2 = 0; h2 < 2; ++h2) _Pragma("unroll") for (int a = 0; a < 2; ++a)  \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(fa[a][h2][ia], fb[b][h2][ib], acc[a][b]);
      UAVGNN_X3_TERM(0, 2) UAVGNN_X3_TERM(2, 0) UAVGNN_X3_TERM(1, 1) UAVGNN_X3_TERM(0, 1) UAVGNN_X3_TERM(1, 0) UAVGNN_X3_TERM(0, 0)
#undef UAVGNN_X3_TERM
    }
  }
  // D layout of a 32 x 32 tile: lane l holds column l % 32, register i holds row 8 (i / 4) + 4 (l / 32) + i % 4
  float* __restrict__ Ps = P + static_cast<size_t>(s) * Mo * Ko;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + wn + b * 32 + l32;
    if (col >= Ko) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm + a * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
        if (row < Mo) {
          float* p = Ps + static_cast<size_t>(row) * Ko + col;
          *p = ACC ? *p + acc[a][b][i] : acc[a][b][i];
        }
      }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_gemm_tn_x3_chunks(long long n_rows, int Mo, int Ko) {
  // row chunks S such that (output tiles) x S fills 256 CUs x 2 resident workgroups, chunks of at least 256 rows
  if (n_rows <= 0 || Mo <= 0 || Ko <= 0) return 0;
  const int tiles = ((Mo + BM - 1) / BM) * ((Ko + BN - 1) / BN);
  long long S = 512 / tiles;
  if (S < 1) S = 1;
  if (S > 256) S = 256;                      // few-tile outputs (a 96- or 9-row weight): enough chunks to fill the chip
#if GEMM_TN_XCD_ORDER
  if (S >= 8) S = (S + 7) / 8 * 8;           // whole rounds over the 8 XCDs (the kernel's XCD-local tile order relies on it)
#endif
  const long long max_s = (n_rows + 255) / 256;   // clamped LAST: no empty row chunks, no partial buffers larger than the rows justify
  if (S > max_s) S = max_s;
  return static_cast<int>(S);
}

extern "C" int uavgnn_gemm_tn_x3(const float* dY, int ldy, int Mo, const float* X, int ldx, int Ko, long long n_rows,
                                 float* partials, int S, int accumulate, uavgnn_stream_t stream) {
  if (!dY || !X || !partials || Mo <= 0 || Ko <= 0 || n_rows <= 0 || S <= 0 || ldy < Mo || ldx < Ko) return UAVGNN_EINVAL;
  if (reinterpret_cast<uintptr_t>(partials) & 3) return UAVGNN_EUNSUPPORTED;
  long long chunk = (n_rows + S - 1) / S;
  chunk = (chunk + BK - 1) / BK * BK;                    // whole 32-row slices; the last chunk may be shorter (or empty)
  const int col_blocks = (Ko + BN - 1) / BN, row_blocks = (Mo + BM - 1) / BM;
  const dim3 grid(row_blocks * col_blocks, S), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (accumulate)
    hipLaunchKernelGGL(gemm_tn_x3_kernel<true>, grid, block, 0, st, dY, ldy, Mo, X, ldx, Ko, n_rows, chunk, partials, col_blocks);
  else
    hipLaunchKernelGGL(gemm_tn_x3_kernel<false>, grid, block, 0, st, dY, ldy, Mo, X, ldx, Ko, n_rows, chunk, partials, col_blocks);
  return launch_status();
}
