// K3a + K3b in ONE launch: the TarMAC message of /root/reference/algos/madrqn/agents/gnn_agents.py:254-267
//   proj = [x || stopgrad(h)] Wp^T + bp            (f_val | f_sign | f_que stacked: value M | signature K | query K columns)
//   c_v  = sum_{u -> v} softmax_u(<s_u, q_v> / K) val_u        over the `talk` relation
// for batches of SMALL graphs with a uniform number of agents (dgl.batch of per-environment graphs, common.py:45,
// env_wrappers.py:139-154: talk edges never leave their environment).  Rounds 1-4 ran this as two vendor GEMMs (x half,
// h half: 19.7 + 19.3 us at C3) + csrc/talk_attn_env.hip (13.5 us) with the [N, M + 2K] projections bounced through HBM:
// 52 us per recurrent step where the traffic floor (x, h read once: 67 MB) is ~13 us.
//
// Layout.  A wavefront owns 16 consecutive agent rows = 16 / n whole environments (n = 1, 2, 4, 8, 16 agents per graph); a
// workgroup is four such wavefronts (64 rows), two workgroups per CU.  The projection GEMM runs on the bf16 matrix cores with
// fp32 accuracy (bf16x3.h: three-way exact operand splits, six v_mfma_f32_16x16x32_bf16 products per fp32 product):
//   * A operand: lane (j, g) <-> row j of the wavefront, k = 8 g .. 8 g + 7 of the 32-wide K slice: two float4 loads straight
//     from x / h (16 rows x 128 B per wave instruction, whole 128-byte segments), split in registers - no LDS, no barrier, since
//     no other wavefront needs these rows;
//   * B operand: the stacked projection weight as bf16 planes in slice-major tiles (uavgnn_tarmac_msg_prepare, built once per
//     weight version by the caller): 18 KB per K slice, copied to LDS by all four wavefronts (double-buffered, one workgroup
//     barrier per slice), read as ds_read_b128 fragments (XOR swizzle of bf16x3.h: conflict-free);
//   * the 16 x (M + 2K) accumulator tile of a wavefront (+ bias) goes to the wavefront's own LDS region and the per-graph
//     attention of csrc/talk_attn_env.hip runs on it in place (lane <-> edge scores, segment softmax, dense A[n x n] in LDS,
//     c = A V with lane <-> channel) - only wave-level synchronisation after the GEMM loop.
// Outputs: c (to any row stride: the GRU input buffer [x || c] of a training step, or a bare [N, M] for no-grad steps), the
// attention weights per CSC position (training), proj (training: what talk_attn_env_bwd and the projection weight gradients
// read), the x half of [x || c] (training; the rows are in registers anyway).
//
// PLANES (optional, `planes_out`): the same launch hands the GRU cell its GEMM operands [x || c || h] as bf16 planes in the
// tile order the cell's loader copies verbatim - the exact three-way split of every activation is made ONCE here, in a
// bandwidth-bound kernel whose VALU is idle, instead of by each of the four column-block workgroups of the cell
// (csrc/gru_x3.hip spends ~70 VALU + 35 LDS stores per 36 MFMAs on it: DESIGN.md section 5).
#include <math.h>

#include "bf16x3.h"
#include "common.h"
#include "uavgnn_probe.h"

namespace uavgnn {
namespace {

using namespace x3;

constexpr int kMsgWaves = 4;
constexpr int kMsgRows = 16 * kMsgWaves;        // rows per workgroup
constexpr int kMsgThreads = kWave * kMsgWaves;
constexpr int kMaxCols = 128;                   // M + 2K (rounded up to 16) <= 128: at most 8 column tiles per wavefront

__device__ __forceinline__ bf16x8 frag_of(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(bf16x8, u32x4{a, b, c, d});
}

// workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() also drains the global stores
// in flight (s_waitcnt vmcnt(0)) - the plane / x-copy stores of a K slice would be waited for at every slice
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct Planes8 {
  u32x4 p[3];
};

// eight consecutive k of one row -> one 16-byte chunk per split plane
__device__ __forceinline__ Planes8 split8(const float4 lo, const float4 hi) {
  const Split3 a = split_pair(lo.x, lo.y), b = split_pair(lo.z, lo.w), c = split_pair(hi.x, hi.y), d = split_pair(hi.z, hi.w);
  Planes8 r;
  r.p[0] = u32x4{a.h1, b.h1, c.h1, d.h1};
  r.p[1] = u32x4{a.h2, b.h2, c.h2, d.h2};
  r.p[2] = u32x4{a.h3, b.h3, c.h3, d.h3};
  return r;
}

// weight [R, KK] (row stride ld) -> tiles [KK / 32][3 planes][RP rows][4 chunks of 8 bf16], chunk c of row r at r * 4 + (c ^
// swz(r)); rows R .. RP - 1 are zero.  One thread per (slice, row, chunk).  The contraction order inside a 32-wide slice is
// PERMUTED: K group c (what lane group c of the MFMA contracts) holds k = 4c .. 4c+3 and 16+4c .. 16+4c+3 - so that the
// activation loads of the forward kernel (lane (j, g): two float4s of row j) cover 64 CONTIGUOUS bytes per row and instruction
// instead of four 16-byte pieces with 16-byte gaps (the sum over k does not care about the order; both operands agree).
__global__ __launch_bounds__(256) void tarmac_msg_prepare_kernel(const float* __restrict__ W, int ld, int R, int RP, int KK,
                                                                 u32x4* __restrict__ tiles) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int per_slice = RP * 4, nsl = KK / 32;
  if (idx >= nsl * per_slice) return;
  const int s = idx / per_slice, rem = idx - s * per_slice, r = rem >> 2, c = rem & 3;
  Planes8 pl;
  if (r < R) {
    const float* src = W + static_cast<size_t>(r) * ld + 32 * s + 4 * c;      // K group c of a slice: k = 4c .. 4c+3 and 16+4c .. 16+4c+3
    pl = split8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 16));
  } else {
    pl.p[0] = pl.p[1] = pl.p[2] = u32x4{0u, 0u, 0u, 0u};
  }
  u32x4* dst = tiles + static_cast<size_t>(s) * 3 * per_slice + r * 4 + (c ^ swz(r));
#pragma unroll
  for (int p = 0; p < 3; ++p) dst[p * per_slice] = pl.p[p];
}

// Deterministic scatter of per-edge values into a dense [NMAX x NMAX] matrix (see csrc/talk_attn_env.hip: parallel edges of one
// (destination, source) pair are summed in CSC order by the lane of the first of them).
__device__ __forceinline__ void msg_scatter(float* __restrict__ Mx, int nmax, const float* __restrict__ VAL,
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

// CT: column tiles of 16 (M + 2K <= 16 CT).  TRAIN: proj / attention weights / x copy are written.  PLANES: the [x || c || h]
// operand planes of the GRU cell are written.
// The attention treats the 16 rows of a wavefront as ONE graph of 16 nodes (its 16 / n whole graphs side by side: a block-diagonal
// dense A[16 x 16]): one pass of the edge / softmax / scatter / A V sequence per wavefront instead of one per graph - that
// sequence is a chain of LDS round trips whose latency, not its work, is what a wavefront pays.
template <int CT, bool TRAIN, bool PLANES>
__global__ __launch_bounds__(kMsgThreads, CT == 8 ? 2 : 3) void tarmac_msg_fwd_kernel(
    const float* __restrict__ x, int ld_x, const float* __restrict__ h, int ld_h, int N, int H, int n_ag,
    const u32x4* __restrict__ Wt, const float* __restrict__ bias, int M, int K, const int32_t* __restrict__ talk_off,
    const int32_t* __restrict__ talk_src, float scale, float* __restrict__ c_out, int ld_c, float* __restrict__ a_save,
    float* __restrict__ proj_out, int ld_p, float* __restrict__ x_copy, int ld_xc, u32x4* __restrict__ planes_out, int dbg) {
  constexpr int RP = 16 * CT;                       // padded projection columns
  constexpr int BCH = 3 * RP * 4;                   // 16-byte chunks of one weight slice (3 planes)
  constexpr int BPT = (BCH + kMsgThreads - 1) / kMsgThreads;
  constexpr int LDP = RP + 1;                       // odd row stride of the projection tile
  constexpr int NA = 16, EMAX = NA * NA;
  constexpr int kScratch = (NA + 4) + 5 * EMAX;     // OFF (padded to 20 words: SC .. AV stay 16-byte aligned) | SC | AD | SRC | DST | AV
  // ONE LDS region, two lives: the double-buffered weight slices while the GEMM loop runs, then (behind the loop's last barrier)
  // the projection tiles + the attention scratch of the four wavefronts - 45 KB at M + 2K = 96: three workgroups per CU
  constexpr int kLoopBytes = 2 * BCH * 16, kTailBytes = kMsgWaves * (16 * LDP + kScratch) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[kLoopBytes > kTailBytes ? kLoopBytes : kTailBytes];
  u32x4(*sB)[BCH] = reinterpret_cast<u32x4(*)[BCH]>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * kMsgRows + wave * 16;          // first row of this wavefront
  const int row = row0 + j, rowc = min(row, N - 1);
  const int nsx = H / 32, ns = 2 * nsx;                        // K slices of x, of [x || h]
  // lane (j, g) contracts k = 4g .. 4g+3 and 16+4g .. 16+4g+3 of a slice (the weight tiles are laid out to match)
  const float* __restrict__ xr = x + static_cast<size_t>(rowc) * ld_x + 4 * g;
  const float* __restrict__ hr = h + static_cast<size_t>(rowc) * ld_h + 4 * g;

  f32x4 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto a_src = [&](int s) {
    s = min(s, ns - 1);
    return s < nsx ? xr + 32 * s : hr + 32 * (s - nsx);
  };
  u32x4 rb[BPT];
  auto load_b = [&](int s) {
    const u32x4* src = Wt + static_cast<size_t>(min(s, ns - 1)) * BCH;
#pragma unroll
    for (int i = 0; i < BPT; ++i) rb[i] = src[min(tid + kMsgThreads * i, BCH - 1)];
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) sB[buf][min(tid + kMsgThreads * i, BCH - 1)] = rb[i];   // clamped duplicates: same data, same address
  };
  // plane tile of the GRU cell: [row block of 128][slice][plane][128 rows][4 chunks], chunk q of row r (k = 8q .. 8q+7) at r * 4 +
  // (q ^ swz32(r)).  A lane holds two 4-k pieces of a slice: k 4g.. -> chunk g >> 1, half g & 1;  k 16+4g.. -> chunk 2 + (g >> 1)
  const int nsl_cell = nsx + (M + 31) / 32 + nsx;               // slices of [x || c || h]
  const int r128 = row & 127;
  u32x2* const pl_base = PLANES ? reinterpret_cast<u32x2*>(planes_out + (static_cast<size_t>(row >> 7) * nsl_cell * 3) * 512 + r128 * 4) +
                                      (g & 1)
                                : nullptr;
  const int pl_q0 = 2 * ((g >> 1) ^ swz32(r128)), pl_q1 = 2 * ((2 + (g >> 1)) ^ swz32(r128));
  auto store_planes = [&](int cell_slice, const Planes8& pl) {
    if (PLANES && row < N) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x2* d = pl_base + (static_cast<size_t>(cell_slice) * 3 + p) * 1024;
        d[pl_q0] = u32x2{pl.p[p][0], pl.p[p][1]};
        d[pl_q1] = u32x2{pl.p[p][2], pl.p[p][3]};
      }
    }
  };

  // activations FOUR slices ahead in a ring of named registers (the x / h rows stream from HBM: what hides their latency is
  // bytes in flight - 3 workgroups x 4 wavefronts x 4 slices x 2 KB = 96 KB per CU)
  float4 a0_lo, a0_hi, a1_lo, a1_hi, a2_lo, a2_hi, a3_lo, a3_hi;
#define UAVGNN_MSG_LOAD_A(LO, HI, S)                          \
  {                                                           \
    const float* p_ = a_src(S);                               \
    LO = *reinterpret_cast<const float4*>(p_);                \
    HI = *reinterpret_cast<const float4*>(p_ + 16);           \
  }
  // Prologue, ordered so that nothing waits behind the activation stream (loads return in issue order: a wait for one of them
  // waits for every older one): the talk offsets and the first weight slice go out FIRST, then four slices of activations; the
  // dependent loads of the talk relation (sources of up to 256 in-edges: the second of two round trips that would otherwise
  // start after the GEMM loop) follow as soon as the offsets are back.
  const int toff = talk_off[min(row0 + min(lane, 16), N)];
  load_b(0);
  __builtin_amdgcn_sched_barrier(0);
  UAVGNN_MSG_LOAD_A(a0_lo, a0_hi, 0)
  UAVGNN_MSG_LOAD_A(a1_lo, a1_hi, 1)
  UAVGNN_MSG_LOAD_A(a2_lo, a2_hi, 2)
  UAVGNN_MSG_LOAD_A(a3_lo, a3_hi, 3)
  __builtin_amdgcn_sched_barrier(0);
  const int e_lo = __shfl(toff, 0);
  const int E = __shfl(toff, 16) - e_lo;
  const bool ok = E <= EMAX && E >= 0;
  store_b(0);
  load_b(1);
  int fsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fsrc[i] = (ok && lane + kWave * i < E) ? talk_src[e_lo + lane + kWave * i] : 0;
  lds_barrier();

  // six products per tile, smallest first, TWO column tiles at a time: v_mfma_f32_16x16x32_bf16 issues at full rate with two
  // independent accumulators between dependent MFMAs (and with sixteen), at HALF rate or worse with 3 .. 8
  // (profiles/r02_ubench_mfma_bf16.txt: 2355 TFLOP/s at 2, 1038-1272 at 3 / 4 / 8) - the round robin over all six tiles that the
  // first version of this kernel used made its GEMM loop MFMA-bound at ~28 us
#define UAVGNN_MSG_TERM(FA, PB)                                                                                      \
  acc[cp] = mfma(FA, as_frag(sb[(PB) * RP * 4 + (cp * 16 + j) * 4 + (g ^ swz(j))]), acc[cp]);                        \
  acc[cp + 1] = mfma(FA, as_frag(sb[(PB) * RP * 4 + ((cp + 1) * 16 + j) * 4 + (g ^ swz(j))]), acc[cp + 1]);
  // one K slice: split the ring entry, (training) its x copy / (planes) its split planes, 6 CT MFMAs against the weight slice in
  // LDS, reload the entry four slices ahead, move the next weight slice into the other buffer
#define UAVGNN_MSG_STEP(LO, HI, T)                                                                   \
  if ((T) < ns) {                                                                                    \
    const int t_ = (T);                                                                              \
    const Planes8 pa = split8(LO, HI);                                                               \
    if (TRAIN && t_ < nsx && x_copy != nullptr && row < N) {   /* the x half of [x || c] */         \
      float* d = x_copy + static_cast<size_t>(row) * ld_xc + 32 * t_ + 4 * g;                        \
      *reinterpret_cast<float4*>(d) = LO;                                                            \
      *reinterpret_cast<float4*>(d + 16) = HI;                                                       \
    }                                                                                                \
    store_planes(t_ < nsx ? t_ : t_ - nsx + nsl_cell - nsx, pa);                                     \
    if (!(dbg & 4)) UAVGNN_MSG_LOAD_A(LO, HI, t_ + 4)                                                \
    const bf16x8 fa0 = as_frag(pa.p[0]), fa1 = as_frag(pa.p[1]), fa2 = as_frag(pa.p[2]);            \
    const u32x4* sb = sB[t_ & 1];                                                                    \
    if (!(dbg & 2)) {                                                                                \
      _Pragma("unroll") for (int cp = 0; cp < CT; cp += 2) {                                         \
        UAVGNN_MSG_TERM(fa0, 2) UAVGNN_MSG_TERM(fa2, 0) UAVGNN_MSG_TERM(fa1, 1) UAVGNN_MSG_TERM(fa0, 1) \
        UAVGNN_MSG_TERM(fa1, 0) UAVGNN_MSG_TERM(fa0, 0)                                              \
      }                                                                                              \
    }                                                                                                \
    if (!(dbg & 1)) {                                                                                \
      store_b((t_ + 1) & 1);   /* its readers passed the barrier of iteration t - 1 */               \
      load_b(t_ + 2);                                                                                \
    }                                                                                                \
    if (!(dbg & 8)) lds_barrier();                                                                   \
  }
  for (int t = 0; t < ns; t += 4) {
    UAVGNN_MSG_STEP(a0_lo, a0_hi, t)
    UAVGNN_MSG_STEP(a1_lo, a1_hi, t + 1)
    UAVGNN_MSG_STEP(a2_lo, a2_hi, t + 2)
    UAVGNN_MSG_STEP(a3_lo, a3_hi, t + 3)
  }
#undef UAVGNN_MSG_STEP
#undef UAVGNN_MSG_TERM
#undef UAVGNN_MSG_LOAD_A

  // ---- projection tile (+ bias) into the wavefront's LDS region: D layout = lane (j, g): column 16 ct + j, rows 4 g .. 4 g + 3
  // (the last barrier of the loop has passed: nobody reads the weight buffers any more - the region becomes the projection tiles
  // and the attention scratch)
  float* __restrict__ P = reinterpret_cast<float*>(smem_raw) + wave * (16 * LDP + kScratch);
  const int ncol = M + 2 * K;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = ct * 16 + j;
    const float b = col < ncol ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) P[(4 * g + i) * LDP + col] = acc[ct][i] + b;
  }
  int* __restrict__ OFF = reinterpret_cast<int*>(P + 16 * LDP);
  float* __restrict__ SC = reinterpret_cast<float*>(OFF + NA + 4);
  float* __restrict__ AD = SC + EMAX;
  int* __restrict__ SRC = reinterpret_cast<int*>(AD + EMAX);
  int* __restrict__ DST = SRC + EMAX;
  float* __restrict__ AV = reinterpret_cast<float*>(DST + EMAX);
  wave_sync_lds();
  if (TRAIN && proj_out != nullptr) {
    for (int r = 0; r < 16; ++r) {
      if (row0 + r >= N) break;
      float* d = proj_out + static_cast<size_t>(row0 + r) * ld_p;
      for (int col = lane; col < ncol; col += kWave) d[col] = P[r * LDP + col];
    }
  }

  // ---- attention over the wavefront's 16 rows as one graph, in place: c overwrites the value columns -------------------------
  if (!ok) {   // more in-edges than the dense matrix holds: fail loudly
    for (int i = 0; i < 16; ++i)
      for (int ch = lane; ch < M; ch += kWave) P[i * LDP + ch] = NAN;
  } else {
    for (int i = lane; i < EMAX; i += kWave) AD[i] = 0.f;
    if (lane <= 16) OFF[lane] = toff - e_lo;
    wave_sync_lds();
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + kWave * i;
      if (e < E) {
        const int u = fsrc[i] - row0;
        bad |= (u < 0) | (u >= 16);
        SRC[e] = u < 0 ? 0 : (u >= 16 ? 15 : u);
        int d = 0;
#pragma unroll
        for (int jj = 1; jj < 16; ++jj) d += e >= OFF[jj];
        DST[e] = d;
      }
    }
    const bool foreign = __any(bad);
    wave_sync_lds();
    if (foreign) {   // an edge enters from outside the tile: the batch is not what the caller said - NaN, never a silent fallback
      for (int i = 0; i < 16; ++i)
        for (int ch = lane; ch < M; ch += kWave) P[i * LDP + ch] = NAN;
    } else {
      for (int e = lane; e < E; e += kWave) {
        const float* __restrict__ sr = P + SRC[e] * LDP + M;
        const float* __restrict__ qr = P + DST[e] * LDP + M + K;
        float a = 0.f;
#pragma unroll 4
        for (int k = 0; k < K; ++k) a = fmaf(sr[k], qr[k], a);
        SC[e] = a * scale;
      }
      wave_sync_lds();
      for (int e = lane; e < E; e += kWave) {   // softmax over the in-edges of the edge's destination
        const int d = DST[e];
        const int j0 = OFF[d], j1 = OFF[d + 1];
        float m = -INFINITY;
#pragma unroll 2
        for (int jj = j0; jj < j1; ++jj) m = fmaxf(m, SC[jj]);
        float den = 0.f;
#pragma unroll 2
        for (int jj = j0; jj < j1; ++jj) den += expf(SC[jj] - m);
        const float a = expf(SC[e] - m) / den;
        if (TRAIN && a_save != nullptr) a_save[e_lo + e] = a;
        AV[e] = a;
      }
      wave_sync_lds();
      msg_scatter(AD, NA, AV, SRC, DST, OFF, E, lane);
      wave_sync_lds();
      // c[r][ch] = sum_t A[r][t] v[t][ch], lane <-> channel, IN PLACE: a lane reads its column of v completely before it writes it
      for (int c0 = 0; c0 < M; c0 += kWave) {
        const int ch = c0 + lane, chc = ch < M ? ch : M - 1;
        float v[NA];
#pragma unroll
        for (int t = 0; t < NA; ++t) v[t] = P[t * LDP + chc];
#pragma unroll 4
        for (int r = 0; r < NA; ++r) {
          float a = 0.f;
#pragma unroll
          for (int t = 0; t < NA; t += 4) {     // a row of A as four 16-byte broadcast reads
            const float4 q = *reinterpret_cast<const float4*>(AD + r * NA + t);
            a = fmaf(q.x, v[t], a);
            a = fmaf(q.y, v[t + 1], a);
            a = fmaf(q.z, v[t + 2], a);
            a = fmaf(q.w, v[t + 3], a);
          }
          if (ch < M) P[r * LDP + ch] = a;
        }
      }
    }
  }
  wave_sync_lds();
  // ---- c: rows to memory, and (PLANES) its split planes as slices nsx .. of the cell operand --------------------------
  for (int r = 0; r < 16; ++r) {
    if (row0 + r >= N) break;
    float* d = c_out + static_cast<size_t>(row0 + r) * ld_c;
    for (int ch = lane; ch < M; ch += kWave) d[ch] = P[r * LDP + ch];
  }
  if (PLANES) {
    for (int cs = 0; cs * 32 < M; ++cs) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ch = 32 * cs + 4 * g + (i & 3) + 16 * (i >> 2);
        v[i] = ch < M ? P[j * LDP + ch] : 0.f;
      }
      store_planes(nsx + cs, split8(float4{v[0], v[1], v[2], v[3]}, float4{v[4], v[5], v[6], v[7]}));
    }
  }
}


// ---- two wavefronts per row tile (split K) ------------------------------------------------------------------------------------
// N / 16 row tiles are two wavefronts per SIMD at C3, each walking a serial chain of 16 slices (load - split - 36 MFMAs - barrier)
// and then the attention: the parts of tools/msg_probe.py's ablation ADD (11 us split + tail, 8 us activation stream, 11.6 us
// weights + MFMAs = 31.5 us).  Here a row tile belongs to a PAIR of wavefronts of one workgroup (eight wavefronts, 64 rows): wave
// `tile` contracts the x half of [x || h] (+ bias), wave `4 + tile` the h half - half the chain each, four wavefronts per SIMD to
// hide each other's waits; a weight stage in LDS holds the x slice t AND the h slice t.  The h partial is added to the tile in
// LDS, then the first wavefront of the pair runs the attention while the second writes `proj` (training).  The sum is
// (x part + bias) + h part - the order of the two-GEMM path of rounds 1-4 -, not the single 16-slice chain of the kernel above.
// RM: max(|x_row|, |c_row|, |h_row|) per agent is written to `row_absmax` - the row scales of the f16x2 GRU cell (csrc/gru_h2.hip); every
// element of x and h passes through this kernel's registers anyway and c is made here (order-independent: deterministic).
template <int CT, bool TRAIN, bool RM = false>
__global__ __launch_bounds__(2 * kMsgThreads, 4) void tarmac_msg_fwd_k2_kernel(
    const float* __restrict__ x, int ld_x, const float* __restrict__ h, int ld_h, int N, int H, int n_ag,
    const u32x4* __restrict__ Wt, const float* __restrict__ bias, int M, int K, const int32_t* __restrict__ talk_off,
    const int32_t* __restrict__ talk_src, float scale, float* __restrict__ c_out, int ld_c, float* __restrict__ a_save,
    float* __restrict__ proj_out, int ld_p, float* __restrict__ x_copy, int ld_xc, float* __restrict__ row_absmax = nullptr) {
  constexpr int kThreads = 2 * kMsgThreads;
  __shared__ float sRowMax[kMsgWaves][2][16];   // RM: per row tile, max |x| (first wavefront of the pair) and max |h| (second)
  float amax = 0.f;
  constexpr int RP = 16 * CT;
  constexpr int BCH = 3 * RP * 4;                   // 16-byte chunks of one weight slice (3 planes)
  constexpr int BPT = (2 * BCH + kThreads - 1) / kThreads;
  constexpr int LDP = RP + 1;
  constexpr int NA = 16, EMAX = NA * NA;
  constexpr int kScratch = (NA + 4) + 5 * EMAX;
  constexpr int kLoopBytes = 2 * 2 * BCH * 16, kTailBytes = kMsgWaves * (16 * LDP + kScratch) * 4;   // [stage][x | h][slice]
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[kLoopBytes > kTailBytes ? kLoopBytes : kTailBytes];
  u32x4(*sB)[2 * BCH] = reinterpret_cast<u32x4(*)[2 * BCH]>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = wave & 3, half = wave >> 2;                 // half 0: x slices, half 1: h slices
  const int j = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * kMsgRows + tile * 16;
  const int row = row0 + j, rowc = min(row, N - 1);
  const int nsx = H / 32;
  const float* __restrict__ ar = (half ? h + static_cast<size_t>(rowc) * ld_h : x + static_cast<size_t>(rowc) * ld_x) + 4 * g;

  f32x4 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight stage t = tiles of slice t (x half) and of slice nsx + t (h half), copied by all eight wavefronts
  u32x4 rb[BPT];
  auto load_b = [&](int t) {
    t = min(t, nsx - 1);
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int q = min(tid + kThreads * i, 2 * BCH - 1);
      rb[i] = Wt[static_cast<size_t>(q < BCH ? t : nsx + t) * BCH + (q < BCH ? q : q - BCH)];
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) sB[buf][min(tid + kThreads * i, 2 * BCH - 1)] = rb[i];
  };
  float4 a0_lo, a0_hi, a1_lo, a1_hi;      // two slices ahead: four wavefronts per SIMD cover the rest of the latency
#define UAVGNN_MSG_LOAD_A(LO, HI, S)                              \
  {                                                               \
    const float* p_ = ar + 32 * min((S), nsx - 1);                \
    LO = *reinterpret_cast<const float4*>(p_);                    \
    HI = *reinterpret_cast<const float4*>(p_ + 16);               \
  }
  const int toff = half == 0 ? talk_off[min(row0 + min(lane, 16), N)] : 0;
  load_b(0);
  __builtin_amdgcn_sched_barrier(0);
  UAVGNN_MSG_LOAD_A(a0_lo, a0_hi, 0)
  UAVGNN_MSG_LOAD_A(a1_lo, a1_hi, 1)
  __builtin_amdgcn_sched_barrier(0);
  const int e_lo = __shfl(toff, 0);
  const int E = __shfl(toff, 16) - e_lo;
  const bool ok = E <= EMAX && E >= 0;
  store_b(0);
  load_b(1);
  // the sources of the tile's first 256 edges: requested here and carried through the GEMM loop - except by the RM instantiations, whose
  // running row maximum takes the kernel over 128 VGPRs (20 bytes of scratch per lane): they request them behind the loop, where the two
  // LDS passes of the pair's tile cover the round trip
  int fsrc[4];
  auto load_fsrc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) fsrc[i] = (half == 0 && ok && lane + kWave * i < E) ? talk_src[e_lo + lane + kWave * i] : 0;
  };
  if (!RM) load_fsrc();
  lds_barrier();

#define UAVGNN_MSG_TERM(FA, PB)                                                                                      \
  acc[cp] = mfma(FA, as_frag(sb[(PB) * RP * 4 + (cp * 16 + j) * 4 + (g ^ swz(j))]), acc[cp]);                        \
  acc[cp + 1] = mfma(FA, as_frag(sb[(PB) * RP * 4 + ((cp + 1) * 16 + j) * 4 + (g ^ swz(j))]), acc[cp + 1]);
#define UAVGNN_MSG_STEP(LO, HI, T)                                                                   \
  if ((T) < nsx) {                                                                                   \
    const int t_ = (T);                                                                              \
    const Planes8 pa = split8(LO, HI);                                                               \
    if (RM) {                                                                                        \
      amax = fmaxf(fmaxf(amax, fabsf(LO.x)), fmaxf(fabsf(LO.y), fabsf(LO.z)));                        \
      amax = fmaxf(fmaxf(amax, fabsf(LO.w)), fmaxf(fabsf(HI.x), fabsf(HI.y)));                        \
      amax = fmaxf(fmaxf(amax, fabsf(HI.z)), fabsf(HI.w));                                           \
    }                                                                                                \
    if (TRAIN && half == 0 && x_copy != nullptr && row < N) {   /* the x half of [x || c] */        \
      float* d = x_copy + static_cast<size_t>(row) * ld_xc + 32 * t_ + 4 * g;                        \
      *reinterpret_cast<float4*>(d) = LO;                                                            \
      *reinterpret_cast<float4*>(d + 16) = HI;                                                       \
    }                                                                                                \
    UAVGNN_MSG_LOAD_A(LO, HI, t_ + 2)                                                                \
    const bf16x8 fa0 = as_frag(pa.p[0]), fa1 = as_frag(pa.p[1]), fa2 = as_frag(pa.p[2]);            \
    const u32x4* sb = sB[t_ & 1] + half * BCH;                                                       \
    _Pragma("unroll") for (int cp = 0; cp < CT; cp += 2) {                                           \
      UAVGNN_MSG_TERM(fa0, 2) UAVGNN_MSG_TERM(fa2, 0) UAVGNN_MSG_TERM(fa1, 1) UAVGNN_MSG_TERM(fa0, 1) \
      UAVGNN_MSG_TERM(fa1, 0) UAVGNN_MSG_TERM(fa0, 0)                                                \
    }                                                                                                \
    store_b((t_ + 1) & 1);   /* its readers passed the barrier of iteration t - 1 */                 \
    load_b(t_ + 2);                                                                                  \
    lds_barrier();                                                                                   \
  }
  for (int t = 0; t < nsx; t += 2) {
    UAVGNN_MSG_STEP(a0_lo, a0_hi, t)
    UAVGNN_MSG_STEP(a1_lo, a1_hi, t + 1)
  }
#undef UAVGNN_MSG_STEP
#undef UAVGNN_MSG_TERM
#undef UAVGNN_MSG_LOAD_A

  // ---- the pair's tile in LDS: (x part + bias) by the first wavefront, + h part by the second ------------------------------
  float* __restrict__ P = reinterpret_cast<float*>(smem_raw) + tile * (16 * LDP + kScratch);
  const int ncol = M + 2 * K;
  if (RM) load_fsrc();
  if (RM) {   // lane (j, g) holds the maximum over its k of row j: combine the four lane groups
    amax = fmaxf(amax, __shfl_xor(amax, 16));
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    if (g == 0) sRowMax[tile][half][j] = amax;
  }
  if (half == 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = ct * 16 + j;
      const float b = col < ncol ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) P[(4 * g + i) * LDP + col] = acc[ct][i] + b;
    }
  }
  lds_barrier();
  if (half == 1) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = ct * 16 + j;
#pragma unroll
      for (int i = 0; i < 4; ++i) P[(4 * g + i) * LDP + col] += acc[ct][i];
    }
  }
  lds_barrier();
  int* __restrict__ OFF = reinterpret_cast<int*>(P + 16 * LDP);
  float* __restrict__ SC = reinterpret_cast<float*>(OFF + NA + 4);
  float* __restrict__ AD = SC + EMAX;
  int* __restrict__ SRC = reinterpret_cast<int*>(AD + EMAX);
  int* __restrict__ DST = SRC + EMAX;
  float* __restrict__ AV = reinterpret_cast<float*>(DST + EMAX);
  if (half == 1) {
    // the second wavefront of the pair: proj rows to memory (training) while the first one computes the attention weights - the
    // value columns are overwritten only behind the next barrier
    if (TRAIN && proj_out != nullptr) {
      for (int r = 0; r < 16; ++r) {
        if (row0 + r >= N) break;
        float* d = proj_out + static_cast<size_t>(row0 + r) * ld_p;
        for (int col = lane; col < ncol; col += kWave) d[col] = P[r * LDP + col];
      }
    }
    lds_barrier();
    return;
  }
  // ---- attention over the tile's 16 rows as one graph (the first wavefront of the pair), as in the kernel above ----------------
  bool poison = !ok;
  if (ok) {
    for (int i = lane; i < EMAX; i += kWave) AD[i] = 0.f;
    if (lane <= 16) OFF[lane] = toff - e_lo;
    wave_sync_lds();
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + kWave * i;
      if (e < E) {
        const int u = fsrc[i] - row0;
        bad |= (u < 0) | (u >= 16);
        SRC[e] = u < 0 ? 0 : (u >= 16 ? 15 : u);
        int d = 0;
#pragma unroll
        for (int jj = 1; jj < 16; ++jj) d += e >= OFF[jj];
        DST[e] = d;
      }
    }
    poison = __any(bad);
    wave_sync_lds();
    if (!poison) {
      for (int e = lane; e < E; e += kWave) {
        const float* __restrict__ sr = P + SRC[e] * LDP + M;
        const float* __restrict__ qr = P + DST[e] * LDP + M + K;
        float a = 0.f;
#pragma unroll 4
        for (int k = 0; k < K; ++k) a = fmaf(sr[k], qr[k], a);
        SC[e] = a * scale;
      }
      wave_sync_lds();
      for (int e = lane; e < E; e += kWave) {
        const int d = DST[e];
        const int j0 = OFF[d], j1 = OFF[d + 1];
        float m = -INFINITY;
#pragma unroll 2
        for (int jj = j0; jj < j1; ++jj) m = fmaxf(m, SC[jj]);
        float den = 0.f;
#pragma unroll 2
        for (int jj = j0; jj < j1; ++jj) den += expf(SC[jj] - m);
        const float a = expf(SC[e] - m) / den;
        if (TRAIN && a_save != nullptr) a_save[e_lo + e] = a;
        AV[e] = a;
      }
      wave_sync_lds();
      msg_scatter(AD, NA, AV, SRC, DST, OFF, E, lane);
      wave_sync_lds();
    }
  }
  lds_barrier();          // the partner has read the value columns (proj): they may be overwritten now
  if (poison) {           // more in-edges than the dense matrix holds, or an edge from outside the tile: fail loudly
    for (int i = 0; i < 16; ++i)
      for (int ch = lane; ch < M; ch += kWave) P[i * LDP + ch] = NAN;
  } else {
    for (int c0 = 0; c0 < M; c0 += kWave) {
      const int ch = c0 + lane, chc = ch < M ? ch : M - 1;
      float v[NA];
#pragma unroll
      for (int t = 0; t < NA; ++t) v[t] = P[t * LDP + chc];
#pragma unroll 4
      for (int r = 0; r < NA; ++r) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < NA; t += 4) {     // a row of A as four 16-byte broadcast reads
          const float4 q = *reinterpret_cast<const float4*>(AD + r * NA + t);
          a = fmaf(q.x, v[t], a);
          a = fmaf(q.y, v[t + 1], a);
          a = fmaf(q.z, v[t + 2], a);
          a = fmaf(q.w, v[t + 3], a);
        }
        if (ch < M) P[r * LDP + ch] = a;
      }
    }
  }
  wave_sync_lds();
  for (int r = 0; r < 16; ++r) {
    if (row0 + r >= N) break;
    float* d = c_out + static_cast<size_t>(row0 + r) * ld_c;
    for (int ch = lane; ch < M; ch += kWave) d[ch] = P[r * LDP + ch];
  }
  if (RM) {   // lane (part g, row j): a quarter of row j's message, then the four parts and the maxima of x and h (NaN messages of a
    float cm = 0.f;   // poisoned tile: fmaxf drops them, the cell turns the NaN elements themselves into NaN outputs)
    for (int ch = g; ch < M; ch += 4) cm = fmaxf(cm, fabsf(P[j * LDP + ch]));
    cm = fmaxf(cm, __shfl_xor(cm, 16));
    cm = fmaxf(cm, __shfl_xor(cm, 32));
    if (g == 0 && row < N) row_absmax[row] = fmaxf(cm, fmaxf(sRowMax[tile][0][j], sRowMax[tile][1][j]));
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

// M + 2K <= 128, K <= 64, H a multiple of 32, graphs of n_ag in {1, 2, 4, 8, 16} agents (16 % n_ag == 0)
extern "C" int uavgnn_tarmac_msg_supported(int H, int M, int K, int n_ag) {
  if (H < 32 || (H % 32) || M < 1 || K < 1 || K > 64 || M + 2 * K > kMaxCols) return 0;
  if (n_ag < 1 || n_ag > 16 || (16 % n_ag)) return 0;
  return 1;
}

// projection columns of the instantiation that serves M + 2K columns (4, 6 or 8 column tiles of 16): the row count of a weight tile
static int msg_rows_padded(int M, int K) {
  const int ct = (M + 2 * K + 15) / 16;
  return 16 * (ct <= 4 ? 4 : ct <= 6 ? 6 : 8);
}

extern "C" long long uavgnn_tarmac_msg_weight_bytes(int H, int M, int K) {
  if (H <= 0 || M <= 0 || K <= 0 || M + 2 * K > kMaxCols) return 0;
  return static_cast<long long>(2 * H / 32) * 3 * msg_rows_padded(M, K) * 64;
}

// bytes of the [x || c || h] operand planes for N rows (whole 128-row blocks)
extern "C" long long uavgnn_tarmac_msg_planes_bytes(int N, int H, int M) {
  if (N <= 0 || H <= 0 || M <= 0) return 0;
  const long long nsl = 2 * (H / 32) + (M + 31) / 32;
  return ((static_cast<long long>(N) + 127) / 128) * nsl * 3 * 8192;
}

extern "C" int uavgnn_tarmac_msg_prepare(const float* Wp, int ld, int H, int M, int K, void* tiles, uavgnn_stream_t stream) {
  if (!Wp || !tiles || H <= 0 || M <= 0 || K <= 0 || ld < 2 * H) return UAVGNN_EINVAL;
  if ((H % 32) || (ld & 3) || (reinterpret_cast<uintptr_t>(Wp) & 15) || (reinterpret_cast<uintptr_t>(tiles) & 15) || M + 2 * K > kMaxCols)
    return UAVGNN_EUNSUPPORTED;
  const int R = M + 2 * K, RP = msg_rows_padded(M, K), KK = 2 * H;
  const int total = (KK / 32) * RP * 4;
  hipLaunchKernelGGL(tarmac_msg_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), Wp, ld,
                     R, RP, KK, static_cast<u32x4*>(tiles));
  return launch_status();
}

// c_out [N, M] (row stride ld_c) always; a_save [E], proj_out [N, M + 2K], x_copy [N, H]: training outputs (each may be NULL);
// planes_out: NULL or uavgnn_tarmac_msg_planes_bytes(N, H, M) bytes.  Every graph has exactly n_ag agents (N % n_ag == 0; 16 %
// n_ag == 0, so no graph straddles two 16-row tiles); the rows of a tile have at most 256 in-edges, all from rows of the same tile -
// a violating tile gets NaN messages, never a silent fallback.
static int msg_launch(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles, const float* bias, int M,
                      int K, const int32_t* talk_off, const int32_t* talk_src, float scale, float* c_out, int ld_c, float* a_save,
                      float* proj_out, int ld_p, float* x_copy, int ld_xc, void* planes_out, float* row_absmax, int dbg,
                      uavgnn_stream_t stream);

// dbg: bits 0-3 are timing ablations of tools/msg_probe.py (results are WRONG when any is set): bit 0 no weight-slice traffic
// after the first two slices, bit 1 no MFMAs, bit 2 no activation loads after the first four slices, bit 3 no workgroup barriers;
// bit 4 (16) selects the one-wavefront-per-row-tile kernel where the default is the wavefront-pair kernel (no planes, M + 2K <=
// 96): correct results, the A/B reference
extern "C" int uavgnn_tarmac_msg_fwd_dbg(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles,
                                         const float* bias, int M, int K, const int32_t* talk_off, const int32_t* talk_src,
                                         float scale, float* c_out, int ld_c, float* a_save, float* proj_out, int ld_p,
                                         float* x_copy, int ld_xc, void* planes_out, int dbg, uavgnn_stream_t stream) {
  return msg_launch(x, ld_x, h, ld_h, N, H, n_ag, tiles, bias, M, K, talk_off, talk_src, scale, c_out, ld_c, a_save, proj_out, ld_p, x_copy,
                    ld_xc, planes_out, nullptr, dbg, stream);
}

static int msg_launch(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles, const float* bias, int M,
                      int K, const int32_t* talk_off, const int32_t* talk_src, float scale, float* c_out, int ld_c, float* a_save,
                      float* proj_out, int ld_p, float* x_copy, int ld_xc, void* planes_out, float* row_absmax, int dbg,
                      uavgnn_stream_t stream) {
  if (N < 0 || !x || !h || !tiles || !bias || !talk_off || !c_out || ld_x < H || ld_h < H || ld_c < M) return UAVGNN_EINVAL;
  if (proj_out && ld_p < M + 2 * K) return UAVGNN_EINVAL;
  if (x_copy && ld_xc < H) return UAVGNN_EINVAL;
  if (!uavgnn_tarmac_msg_supported(H, M, K, n_ag) || (N % n_ag) || (ld_x & 3) || (ld_h & 3) || (x_copy && (ld_xc & 3)) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(tiles) |
        reinterpret_cast<uintptr_t>(x_copy) | reinterpret_cast<uintptr_t>(planes_out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const int ct = (M + 2 * K + 15) / 16;
  const bool train = a_save != nullptr || proj_out != nullptr || x_copy != nullptr;
  const bool planes = planes_out != nullptr;
  const dim3 grid((N + kMsgRows - 1) / kMsgRows), block(kMsgThreads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const u32x4* Wt = static_cast<const u32x4*>(tiles);
  u32x4* po = static_cast<u32x4*>(planes_out);
#define UAVGNN_MSG_LAUNCH(NA_, CT_, TR_, PL_)                                                                              \
  hipLaunchKernelGGL((tarmac_msg_fwd_kernel<CT_, TR_, PL_>), grid, block, 0, st, x, ld_x, h, ld_h, N, H, n_ag, Wt, bias, M, \
                     K, talk_off, talk_src, scale, c_out, ld_c, a_save, proj_out, ld_p, x_copy, ld_xc, po, dbg)
#define UAVGNN_MSG_BY_FLAGS(NA_, CT_)                                    \
  {                                                                      \
    if (train && planes) UAVGNN_MSG_LAUNCH(NA_, CT_, true, true);        \
    else if (train) UAVGNN_MSG_LAUNCH(NA_, CT_, true, false);            \
    else if (planes) UAVGNN_MSG_LAUNCH(NA_, CT_, false, true);           \
    else UAVGNN_MSG_LAUNCH(NA_, CT_, false, false);                      \
  }
#define UAVGNN_MSG_BY_CT(NA_)                                            \
  {                                                                      \
    if (ct <= 4) UAVGNN_MSG_BY_FLAGS(NA_, 4)                             \
    else if (ct <= 6) UAVGNN_MSG_BY_FLAGS(NA_, 6)                        \
    else UAVGNN_MSG_BY_FLAGS(NA_, 8)                                     \
  }
  if (row_absmax != nullptr && (planes || ct > 6 || (dbg & 31))) return UAVGNN_EUNSUPPORTED;   // written by the wavefront-pair kernel only
  if (!planes && ct <= 6 && !(dbg & 31)) {
    // two wavefronts per row tile (73 KB of LDS at six column tiles: two workgroups per CU; eight column tiles would be one)
    const dim3 block2(2 * kMsgThreads);
#define UAVGNN_MSG_K2(CT_, TR_)                                                                                                        \
  {                                                                                                                                    \
    if (row_absmax != nullptr)                                                                                                         \
      hipLaunchKernelGGL((tarmac_msg_fwd_k2_kernel<CT_, TR_, true>), grid, block2, 0, st, x, ld_x, h, ld_h, N, H, n_ag, Wt, bias, M, K, \
                         talk_off, talk_src, scale, c_out, ld_c, a_save, proj_out, ld_p, x_copy, ld_xc, row_absmax);                   \
    else                                                                                                                               \
      hipLaunchKernelGGL((tarmac_msg_fwd_k2_kernel<CT_, TR_, false>), grid, block2, 0, st, x, ld_x, h, ld_h, N, H, n_ag, Wt, bias, M, K, \
                         talk_off, talk_src, scale, c_out, ld_c, a_save, proj_out, ld_p, x_copy, ld_xc, nullptr);                      \
  }
    if (ct <= 4) {
      if (train) UAVGNN_MSG_K2(4, true)
      else UAVGNN_MSG_K2(4, false)
    } else {
      if (train) UAVGNN_MSG_K2(6, true)
      else UAVGNN_MSG_K2(6, false)
    }
#undef UAVGNN_MSG_K2
    return launch_status();
  }
  UAVGNN_MSG_BY_CT(16)
#undef UAVGNN_MSG_BY_CT
#undef UAVGNN_MSG_BY_FLAGS
#undef UAVGNN_MSG_LAUNCH
  return launch_status();
}

// ... that also writes row_absmax [N] = max(|x_row|, |c_row|, |h_row|): the row scales of uavgnn_gru_cell_fwd_h2.  M + 2K <= 96, no
// planes_out (UAVGNN_EUNSUPPORTED otherwise: the caller runs uavgnn_tarmac_msg_fwd and the bf16x3 cell)
extern "C" int uavgnn_tarmac_msg_rowmax_supported(int H, int M, int K, int n_ag) {
  return (uavgnn_tarmac_msg_supported(H, M, K, n_ag) && (M + 2 * K + 15) / 16 <= 6) ? 1 : 0;
}

extern "C" int uavgnn_tarmac_msg_fwd_rowmax(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles,
                                            const float* bias, int M, int K, const int32_t* talk_off, const int32_t* talk_src,
                                            float scale, float* c_out, int ld_c, float* a_save, float* proj_out, int ld_p,
                                            float* x_copy, int ld_xc, float* row_absmax, uavgnn_stream_t stream) {
  if (!row_absmax) return UAVGNN_EINVAL;
  return msg_launch(x, ld_x, h, ld_h, N, H, n_ag, tiles, bias, M, K, talk_off, talk_src, scale, c_out, ld_c, a_save, proj_out, ld_p, x_copy,
                    ld_xc, nullptr, row_absmax, 0, stream);
}

extern "C" int uavgnn_tarmac_msg_fwd(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles,
                                     const float* bias, int M, int K, const int32_t* talk_off, const int32_t* talk_src,
                                     float scale, float* c_out, int ld_c, float* a_save, float* proj_out, int ld_p,
                                     float* x_copy, int ld_xc, void* planes_out, uavgnn_stream_t stream) {
  return uavgnn_tarmac_msg_fwd_dbg(x, ld_x, h, ld_h, N, H, n_ag, tiles, bias, M, K, talk_off, talk_src, scale, c_out, ld_c, a_save,
                                   proj_out, ld_p, x_copy, ld_xc, planes_out, 0, stream);
}
