// K4 on the bf16 matrix cores: the whole GRU cell in one kernel (as gru_fused.hip), with the two GEMMs computed as exact
// three-way bf16 splits of the fp32 operands and six bf16 MFMA products per fp32 product (bf16x3.h: fp32-level error, below
// rocBLAS sgemm's on the same data; fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950).
// Replaces nn.GRUCell at /root/reference/algos/madrqn/agents/gnn_agents.py:29,:123,:164,:208,:246,:282 (gate order r, z, n):
//   r = sigma(W_ir i + b_ir + W_hr h + b_hr)   z = sigma(W_iz i + b_iz + W_hz h + b_hz)
//   n = tanh(W_in i + b_in + r (W_hn h + b_hn))   h' = (1 - z) n + z h
//
// Data flow.  uavgnn_gru_split_weights turns W_ih [3H, K_in] and W_hh [3H, H] into bf16 planes [3][3H][K] (one launch per
// forward: the weights of a drop-in module may change behind any cache - `.data` writes do not bump a version counter - and
// the launch costs 3 us).  The cell kernel walks K in 32-wide slices over [x || c] (W_ih rows) and then h (W_hh rows) with
// four accumulator sets r, z, gi_n, gh_n (r and z take both contractions); per slice the activation tile is loaded as fp32,
// split in registers (v_cvt_pk_bf16_f32 + two subtractions per term) and written to LDS as three bf16 planes, the weight
// planes are copied; LDS rows are 64 B with an XOR swizzle (conflict-free ds_read_b128 fragments, no padding).  The column
// blocks of a row block run on the same XCD (blockIdx -> XCD round robin), so the activation tile is fetched from HBM once
// and re-read from that XCD's L2.  Epilogue as gru_fused.hip: biases, gates, h' through an LDS tile (full-row HBM
// accesses), optional pre-activations [N, 4H] for uavgnn_gru_gates_bwd_fused.
//
// gru_cell_fwd_x3w8_kernel (H % 64 == 0; other shapes take the fp32-MFMA cell of gru_fused.hip): 512 threads own 128 agents
// x 64 hidden units; eight wavefronts of 32 agents x 32 units x 4 sets (64 accumulator registers) on
// v_mfma_f32_32x32x16_bf16.  The LDS tiles are double-buffered with ONE barrier per slice, the fragment reads are software-
// pipelined over the two 16-wide halves of a slice, the loads of slice t+2 are in flight while slice t computes.
// Measured (tools/ubench/gru_x3_bench.hip, C3 size; round 2, staging in blocks - round 3's interleaved staging: 163-165 us): 194 us; without any staging (fragment reads + MFMAs + barriers) 161 us;
// prologue + epilogue alone 36 us (one workgroup per CU: nothing overlaps them); the MFMAs alone would take 94 us
// (tools/ubench/mfma_bf16.hip: 18.2 ns per 32x32x16 MFMA per SIMD with random operands).  Variants measured and dropped:
// 128 x 32 tiles with four waves and 16x16x32 MFMAs (200 us: three independent accumulators between dependent 16x16x32
// MFMAs issue at half rate, profiles/r02_ubench_mfma_bf16.txt), the same with 32x32x16 MFMAs (215 us), loads two slices
// ahead through a fully unrolled slice loop (-4 us), a start skew between co-resident workgroups (0), persistent workgroups
// that load the next tile's first slice during the epilogue (207 us), the epilogue's h tile and biases requested in front of the
// slice loop and held in registers (round 4, same-box A/B: 202.5 vs 193.8 us and 175.1 vs 167.2 us - slower: 22 more live registers).
#include <type_traits>

#include "bf16x3.h"
#include "common.h"

namespace uavgnn {
namespace {

using namespace x3;
constexpr int BM = 128, BK = 32;

// gate non-linearities on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each): absolute error <= 2e-7, far inside
// the 1e-5 parity tolerance; tanh as 1 - 2 / (1 + e^{2x}) saturates correctly at both ends (e^{2x} -> inf / 0)
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

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

// products of one fp32 product, smallest first: (a1 b3) (a3 b1) (a2 b2) (a1 b2) (a2 b1) (a1 b1)
#define UAVGNN_X3_FOR_TERMS(M) M(0, 2) M(2, 0) M(1, 1) M(0, 1) M(1, 0) M(0, 0)

// ---- eight wavefronts, 128 agents x 64 hidden units, double-buffered LDS, one barrier per slice ----------------------------
namespace w8 {
constexpr int BJ = 64, NT = 512, ST = 68;
constexpr int PA = BM * 4, PB = 3 * BJ * 4;            // 16-byte chunks per split plane of the A / B tile
constexpr int BUF = 3 * PA + 3 * PB;                   // chunks per buffer (60 KB)
}  // namespace w8

template <bool SAVE, bool IL>
__global__ __launch_bounds__(w8::NT) void gru_cell_fwd_x3w8_kernel(
    const float* __restrict__ inp, int ld_inp, int K1, const float* __restrict__ inp2, int ld_inp2, int K2,
    const float* __restrict__ h, int N, int H,
    const unsigned short* __restrict__ Wih_p, const float* __restrict__ b_ih, const unsigned short* __restrict__ Whh_p,
    const float* __restrict__ b_hh, float* __restrict__ h_out, float* __restrict__ pre, int row_blocks) {
  using namespace w8;
  __shared__ u32x4 smem[2 * BUF];   // buffer b: A planes [3][128][4] then B planes [3][192 = gate * 64 + unit][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int CB = H / BJ;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot / CB) * 8 + xcd, cb = slot - (slot / CB) * CB;
  if (rb >= row_blocks) return;
  const int m0 = rb * BM, j0 = cb * BJ;

  f32x16 acc[4];     // 32 x 32 tile per set: r, z, gi_n, gh_n
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;
  const int l32 = lane & 31, lh = lane >> 5, sw = swz32(l32);
  // A loader: float4 q = tid + 512 i -> row tid / 8 + 64 i, k = 4 (tid % 8); rows past N are clamped (stores are masked)
  const int lr = tid >> 3, c4 = tid & 7;
  const int sa_w = lr * 32 + (((c4 >> 1) ^ swz32(lr)) * 8) + (c4 & 1) * 4;   // in bf16 units inside an A plane
  // B loader: chunk q = tid + 512 i (i < 5, 2304 chunks): plane q / 768, row (q % 768) / 4 = gate * 64 + unit, chunk q % 4
  // Loads walk the K slices through a CURSOR: scalar base pointers of the activation piece / weight-plane set in use, advanced
  // by one slice per load, plus per-lane BYTE offsets of that piece (global_load with an SGPR base + 32-bit VGPR offset: no
  // per-slice address arithmetic on the VALU).  At a piece boundary - [inp (K1 columns) || inp2 (K2 columns)] against W_ih, then
  // h against W_hh; inp2 is the TarMAC step's c without the concatenated [x || c] copy, K2 = 0: one piece - a wave-uniform
  // branch re-bases the cursor and recomputes the lane offsets (twice per kernel).
  unsigned rowa[2], wrow[5];
  int sbw[5];
#pragma unroll
  for (int i = 0; i < 2; ++i) rowa[i] = static_cast<unsigned>(min(m0 + lr + 64 * i, N - 1));
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int q = min(tid + NT * i, 3 * PB - 1), pl = q / PB, rem = q - pl * PB, row = rem >> 2, c = rem & 3;
    wrow[i] = static_cast<unsigned>(pl * 3 * H + (row >> 6) * H + j0 + (row & 63));
    sbw[i] = 3 * PA + pl * PB + row * 4 + (c ^ swz32(row));
  }
  // byte offset of the lane's 8-bf16 chunk inside a weight slice: chunk q % 4 (= tid % 4, except for the clamped i = 4 of tid >= 256)
  const unsigned wc8 = 16u * (tid & 3), wc8_last = 16u * (min(tid + NT * 4, 3 * PB - 1) & 3);
  const int n1 = K1 / BK, n12 = n1 + K2 / BK, ns = n12 + H / BK;   // slices of inp, of [inp || inp2], of everything
  float4 ra[2];
  u32x4 rw[5];
  unsigned oa[2], ow[5];
  const char* __restrict__ Ab = reinterpret_cast<const char*>(inp);
  const char* __restrict__ Wb = reinterpret_cast<const char*>(Wih_p);
  int lt = 0;                                                // the slice the next load fetches
  auto set_a = [&](const float* base, int ld) {
    Ab = reinterpret_cast<const char*>(base);
#pragma unroll
    for (int i = 0; i < 2; ++i) oa[i] = 4u * (rowa[i] * static_cast<unsigned>(ld) + 4u * c4);
  };
  auto set_w = [&](const unsigned short* base, int K) {
    Wb = reinterpret_cast<const char*>(base);
#pragma unroll
    for (int i = 0; i < 5; ++i) ow[i] = 2u * wrow[i] * static_cast<unsigned>(K) + (i == 4 ? wc8_last : wc8);
  };
  set_a(inp, ld_inp);
  set_w(Wih_p, K1 + K2);
#ifndef UAVGNN_X3_DBG_LD
#define UAVGNN_X3_DBG_LD 3   /* timing experiments: bit 0 = activation loads, bit 1 = weight-plane loads inside the slice loop */
#endif
  auto gload_a = [&]() {
    if ((UAVGNN_X3_DBG_LD & 1) || lt < 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(Ab + oa[i]);
    }
  };
  auto gload_w = [&]() {
    if ((UAVGNN_X3_DBG_LD & 2) || lt < 2) {
#pragma unroll
      for (int i = 0; i < 5; ++i) rw[i] = *reinterpret_cast<const u32x4*>(Wb + ow[i]);
    }
  };
  auto advance = [&]() {      // after both loads of slice lt were issued; past the last slice the cursor stays on it
    if (lt + 1 >= ns) return;
    ++lt;
    Ab += 4 * BK;
    Wb += 2 * BK;
    if (lt == n1 && n12 > n1) set_a(inp2, ld_inp2);
    if (lt == n12) {
      set_a(h, H);
      set_w(Whh_p, H);
    }
  };
  auto gload = [&]() {
    gload_a();
    gload_w();
    advance();
  };
  auto lstore_b = [&](int buf) {
    u32x4* sb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 5; ++i) sb[sbw[i]] = rw[i];   // i = 4, tid >= 256: the clamped chunk again - same data, same address
  };
  auto lstore_a = [&](int buf, int i) {
    unsigned short* sa = reinterpret_cast<unsigned short*>(smem + buf * BUF) + sa_w;
    stage4(sa + 64 * i * 32, PA * 8, ra[i]);
  };
  auto lstore = [&](int buf) {
    lstore_a(buf, 0);
    lstore_a(buf, 1);
    lstore_b(buf);
  };
  // One K slice = two 16-wide halves of v_mfma_f32_32x32x16_bf16 (lane -> row lane % 32, 8 consecutive k at 8 (lane / 32) of
  // the half): per half and gate one 32 x 32 tile and six products; the three gates interleave so that dependent MFMAs are
  // three apart.  The 32x32 shape is chosen for its issue behaviour: 16x16x32 bf16 MFMAs with 3-8 independent accumulators
  // between dependent ones run at HALF rate or worse (profiles/r02_ubench_mfma_bf16.txt: 457-1075 TFLOP/s vs 2140 for
  // 32x32x16 at any distance).
  struct Half {
    bf16x8 a[3], b[3][3];   // [plane], [gate][plane]
  };
#define UAVGNN_X3_W8_READ(F, buf, kh)                                                                              \
  {                                                                                                                \
    const u32x4* sb = smem + (buf) * BUF;                                                                          \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) F.a[pl] = as_frag(sb[pl * PA + (wm + l32) * 4 + ((2 * (kh) + lh) ^ sw)]); \
    _Pragma("unroll") for (int gate = 0; gate < 3; ++gate) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)        \
        F.b[gate][pl] = as_frag(sb[3 * PA + pl * PB + (gate * BJ + wc + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);       \
  }
#define UAVGNN_X3_W8_TERM(ia, ib)                  \
  acc[0] = mfma32(F.a[ia], F.b[0][ib], acc[0]);    \
  acc[1] = mfma32(F.a[ia], F.b[1][ib], acc[1]);    \
  acc[NSET] = mfma32(F.a[ia], F.b[2][ib], acc[NSET]);
#define UAVGNN_X3_W8_MFMA(F_, NSET_)               \
  {                                                \
    constexpr int NSET = NSET_;                    \
    const Half& F = F_;                            \
    UAVGNN_X3_FOR_TERMS(UAVGNN_X3_W8_TERM)         \
  }

  gload();
  lstore(0);
  gload();
  __syncthreads();
  // Software pipeline over the halves: the fragment reads of a half are issued one MFMA group (18 MFMAs = 576 cycles) before
  // their use - f1 = (slice t, second half) under the MFMAs of f0, f0 = (slice t + 1, first half) right after the barrier
  // under the MFMAs of f1.  Iteration t also stages slice t + 1 into the other buffer (its readers passed the barrier of
  // iteration t - 1) and starts the loads of slice t + 2; the tail re-stages / re-loads the last slice (unconditional).
  // Two schedules of the staging (template parameter IL, UAVGNN_GRU_STAGING_BLOCKS of uavgnn_gru_cell_fwd_x3_opts; bit-identical results):
  //   blocks      - waves w and w + 4 share a SIMD: the first four stage BEFORE their first MFMA group, the last four AFTER it;
  //   interleaved - every wave stages INSIDE its first MFMA group (the default; see UAVGNN_X3_W8_STEP_IL).
  // sched_barrier pins the order (left alone, the compiler sinks the global loads below the MFMAs - prefetch distance zero -
  // and issues the LDS reads right before their MFMAs).
  const bool early = wave < 4;
  Half f0, f1;
  UAVGNN_X3_W8_READ(f0, 0, 0)
  int t = 0;
#ifndef UAVGNN_X3_DBG
#define UAVGNN_X3_DBG 0   /* timing experiments of tools/ubench/gru_x3_bench.hip: 1 = no global loads in the loop, 2 = no staging */
#endif
#define UAVGNN_X3_W8_STEP_BLK(NSET)                        \
  UAVGNN_X3_W8_READ(f1, t & 1, 1)                          \
  if (early) {                                             \
    if (UAVGNN_X3_DBG < 2) lstore((t + 1) & 1);            \
    if (UAVGNN_X3_DBG < 1) gload();                        \
  }                                                        \
  __builtin_amdgcn_sched_barrier(0);                       \
  UAVGNN_X3_W8_MFMA(f0, NSET)                              \
  __builtin_amdgcn_sched_barrier(0);                       \
  if (!early) {                                            \
    if (UAVGNN_X3_DBG < 2) lstore((t + 1) & 1);            \
    if (UAVGNN_X3_DBG < 1) gload();                        \
  }                                                        \
  __syncthreads();                                         \
  UAVGNN_X3_W8_READ(f0, (t + 1) & 1, 0)                    \
  __builtin_amdgcn_sched_barrier(0);                       \
  UAVGNN_X3_W8_MFMA(f1, NSET)                              \
  __builtin_amdgcn_sched_barrier(0);
  // IL: the staging of slice t + 1 and the loads of slice t + 2 are INTERLEAVED with the first MFMA group in program order
  // (sched_group_barrier; every wave stages there).  An independent VALU / LDS / memory instruction issues in the shadow of an
  // executing MFMA only when it FOLLOWS it in the instruction stream; as a block in front of the MFMAs its issue time adds to
  // theirs (tools/ubench/mfma_bf16.hip: 19.1 -> 15.8 ns per MFMA with ~90 staging VALU per 36 MFMAs).  The weight-plane loads
  // go out right behind the LDS stores of their registers, the activation loads behind the splits.
#define UAVGNN_X3_W8_STEP_IL(NSET_)                        \
  {                                                        \
    constexpr int NSET = NSET_;                            \
    UAVGNN_X3_W8_READ(f1, t & 1, 1)                        \
    __builtin_amdgcn_sched_barrier(0);                     \
    const Half& F = f0;                                    \
    if (UAVGNN_X3_DBG < 2) lstore_b((t + 1) & 1);          \
    if (UAVGNN_X3_DBG < 1) gload_w();                      \
    UAVGNN_X3_W8_TERM(0, 2)                                \
    _Pragma("unroll") for (int sg = 0; sg < 3; ++sg) {     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   \
    }                                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    if (UAVGNN_X3_DBG < 2) { lstore_a((t + 1) & 1, 0); lstore_a((t + 1) & 1, 1); } \
    UAVGNN_X3_W8_TERM(2, 0) UAVGNN_X3_W8_TERM(1, 1) UAVGNN_X3_W8_TERM(0, 1) UAVGNN_X3_W8_TERM(1, 0) \
    _Pragma("unroll") for (int sg = 0; sg < 12; ++sg) {    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   \
    }                                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    if (UAVGNN_X3_DBG < 1) gload_a();                      \
    UAVGNN_X3_W8_TERM(0, 0)                                \
    _Pragma("unroll") for (int sg = 0; sg < 3; ++sg) {     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   \
    }                                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    if (UAVGNN_X3_DBG < 1) advance();                      \
  }                                                        \
  __syncthreads();                                         \
  UAVGNN_X3_W8_READ(f0, (t + 1) & 1, 0)                    \
  __builtin_amdgcn_sched_barrier(0);                       \
  UAVGNN_X3_W8_MFMA(f1, NSET_)                             \
  __builtin_amdgcn_sched_barrier(0);
  if (UAVGNN_X3_DBG == 3) t = ns;   /* timing experiment: prologue + epilogue only */
  if (IL) {
    for (; t < n12; ++t) { UAVGNN_X3_W8_STEP_IL(2) }
    for (; t < ns; ++t) { UAVGNN_X3_W8_STEP_IL(3) }
  } else {
    for (; t < n12; ++t) { UAVGNN_X3_W8_STEP_BLK(2) }
    for (; t < ns; ++t) { UAVGNN_X3_W8_STEP_BLK(3) }
  }
#undef UAVGNN_X3_W8_STEP_IL
#undef UAVGNN_X3_W8_STEP_BLK
#undef UAVGNN_X3_W8_MFMA
#undef UAVGNN_X3_W8_READ
  __syncthreads();   // the last iteration's read of the stale buffer must not race the epilogue's tile
#undef UAVGNN_X3_W8_TERM
  // ---- epilogue on the D layout: lane (g, j) holds rows 4g..4g+3 of a tile, one hidden unit ------------------------------
  // (the last barrier of the loop has passed: nobody reads the buffers any more)
  float* sH = reinterpret_cast<float*>(smem);             // [128][ST] fp32 tile: h in, h' out, 16-byte row-contiguous HBM accesses
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + NT * q, row = idx >> 4, cc = idx & 15;
    *reinterpret_cast<float4*>(sH + row * ST + 4 * cc) =
        *reinterpret_cast<const float4*>(h + static_cast<size_t>(min(m0 + row, N - 1)) * H + j0 + 4 * cc);
  }
  __syncthreads();
  // D layout of the 32 x 32 tile: lane l holds column l % 32, register i holds row 8 (i / 4) + 4 (l / 32) + i % 4
  const int c = j0 + wc + l32;
  const float b_r = b_ih[c] + b_hh[c], b_z = b_ih[H + c] + b_hh[H + c], b_in = b_ih[2 * H + c], b_hn = b_hh[2 * H + c];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int lrow = wm + 8 * (i >> 2) + 4 * lh + (i & 3);
    const int row = m0 + lrow;
    const float pr = acc[0][i] + b_r, pz = acc[1][i] + b_z, gin = acc[2][i] + b_in, ghn = acc[3][i] + b_hn;
    const float rr = sigmoidf_(pr), zz = sigmoidf_(pz);
    const float nn = tanhf_(fmaf(rr, ghn, gin));
    float* hp = sH + lrow * ST + wc + l32;
    *hp = fmaf(zz, *hp - nn, nn);                          // every element of the tile has exactly one owner lane
    if (SAVE && row < N) {
      float* p = pre + static_cast<size_t>(row) * 4 * H + c;
      p[0] = pr;
      p[H] = pz;
      p[2 * H] = gin;
      p[3 * H] = ghn;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + NT * q, row = idx >> 4, cc = idx & 15;
    if (m0 + row < N)
      *reinterpret_cast<float4*>(h_out + static_cast<size_t>(m0 + row) * H + j0 + 4 * cc) =
          *reinterpret_cast<const float4*>(sH + row * ST + 4 * cc);
  }
}


}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_gru_cell_x3_supported(int K_in, int H) {
  return (K_in >= BK && K_in % BK == 0 && H >= w8::BJ && H % w8::BJ == 0) ? 1 : 0;
}

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

// flags: UAVGNN_GRU_STAGING_BLOCKS = the round-2 schedule (staging of a slice as a block in front of / behind the first MFMA
// group) instead of the interleaved one - the A/B of tools/gru_probe.py, bit-identical results.  Per call: no global state.
extern "C" int uavgnn_gru_cell_fwd_x3_opts(const float* inp, int ld_inp, int K1, const float* inp2, int ld_inp2, int K2,
                                           const float* h, int N, int H, const void* planes, const float* b_ih,
                                           const float* b_hh, float* h_out, float* pre_save, int flags,
                                           uavgnn_stream_t stream) {
  const int K_in = K1 + K2;
  if (N < 0 || !inp || !h || !planes || !b_ih || !b_hh || !h_out || ld_inp < K1 || K2 < 0 || (K2 > 0 && (!inp2 || ld_inp2 < K2)))
    return UAVGNN_EINVAL;
  if (K2 == 0) {
    inp2 = inp;
    ld_inp2 = ld_inp;
  }
  if (!uavgnn_gru_cell_x3_supported(K_in, H) || K1 < BK || (K1 % BK) || (K2 % BK) || (ld_inp & 3) || (ld_inp2 & 3) ||
      ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(inp2) | reinterpret_cast<uintptr_t>(h) |
        reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(h_out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  // the kernel addresses its operands by 32-bit BYTE offsets from the base pointers (global_load with an SGPR base)
  const long long ld_max = ld_inp > ld_inp2 ? (ld_inp > H ? ld_inp : H) : (ld_inp2 > H ? ld_inp2 : H);
  if (4LL * N * ld_max >= (1LL << 32) || 18LL * H * (K_in > H ? K_in : H) >= (1LL << 32)) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const unsigned short* p0 = static_cast<const unsigned short*>(planes);
  const unsigned short* p1 = p0 + 9LL * H * K_in;
  const int row_blocks = (N + BM - 1) / BM, rb8 = ((row_blocks + 7) / 8) * 8;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define UAVGNN_X3_LAUNCH(KERNEL, BJ_, NT_)                                                                                  \
  hipLaunchKernelGGL(KERNEL, dim3(rb8 * (H / (BJ_))), dim3(NT_), 0, st, inp, ld_inp, K1, inp2, ld_inp2, K2, h, N, H, p0, b_ih, \
                     p1, b_hh, h_out, pre_save, row_blocks)
  if (!(flags & UAVGNN_GRU_STAGING_BLOCKS)) {
    if (pre_save != nullptr) UAVGNN_X3_LAUNCH((gru_cell_fwd_x3w8_kernel<true, true>), w8::BJ, w8::NT);
    else UAVGNN_X3_LAUNCH((gru_cell_fwd_x3w8_kernel<false, true>), w8::BJ, w8::NT);
  } else {
    if (pre_save != nullptr) UAVGNN_X3_LAUNCH((gru_cell_fwd_x3w8_kernel<true, false>), w8::BJ, w8::NT);
    else UAVGNN_X3_LAUNCH((gru_cell_fwd_x3w8_kernel<false, false>), w8::BJ, w8::NT);
  }
#undef UAVGNN_X3_LAUNCH
  return launch_status();
}

extern "C" int uavgnn_gru_cell_fwd_x3_cat(const float* inp, int ld_inp, int K1, const float* inp2, int ld_inp2, int K2,
                                          const float* h, int N, int H, const void* planes, const float* b_ih,
                                          const float* b_hh, float* h_out, float* pre_save, uavgnn_stream_t stream) {
  return uavgnn_gru_cell_fwd_x3_opts(inp, ld_inp, K1, inp2, ld_inp2, K2, h, N, H, planes, b_ih, b_hh, h_out, pre_save, 0, stream);
}

extern "C" int uavgnn_gru_cell_fwd_x3(const float* inp, int ld_inp, int K_in, const float* h, int N, int H,
                                      const void* planes, const float* b_ih, const float* b_hh, float* h_out,
                                      float* pre_save, uavgnn_stream_t stream) {
  return uavgnn_gru_cell_fwd_x3_cat(inp, ld_inp, K_in, nullptr, 0, 0, h, N, H, planes, b_ih, b_hh, h_out, pre_save, stream);
}
