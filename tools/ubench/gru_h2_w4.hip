// EXPERIMENT, not part of libuavgnn.so (tools/h2_ablate.py --ab): csrc/gru_h2.hip with workgroups of FOUR wavefronts on 64 x 64 tiles, two
// workgroups per CU (64 KB of LDS each) - two independent barrier domains per CU instead of one, at 1.6 x the L2 -> LDS bytes per row.
// K4 on the f16 matrix cores with an exactly scaled TWO-term split ("f16x2"): the whole GRU cell in one kernel, as gru_x3.hip, at
// HALF the matrix-core work - three f16 products per fp32 product instead of bf16x3's six.
// Replaces nn.GRUCell at /root/reference/algos/madrqn/agents/gnn_agents.py:246 (TarMAC's f_udt; gate order r, z, n):
//   r = sigma(W_ir i + b_ir + W_hr h + b_hr)   z = sigma(W_iz i + b_iz + W_hz h + b_hz)
//   n = tanh(W_in i + b_in + r (W_hn h + b_hn))   h' = (1 - z) n + z h
//
// Why.  Round 5's ablation of the bf16x3 cell ("three designs, one plateau"): at the package power limit the 36
// MFMAs per slice and wavefront alone take 105 of the kernel's 161 us and nothing overlaps them - the lever is the NUMBER of MFMAs.
//
// Arithmetic.  fp32 in / out / accumulate.  Every row of the activation operand A = [x || c || h] and every output unit's row of the
// stacked weights [W_ih | W_hh] gets a POWER-OF-TWO scale that puts its largest magnitude into [2^14, 2^15) (an exponent add: exact);
// the scaled value is split as hi = rn_f16(v), lo = rn_f16(v - hi).  v - hi is exact in fp32 and has at most 13 significant bits, so
// hi + lo reproduces v to <= 2^-23 |v| (exactly for three elements in four); elements below 2^-17 of their row's maximum lose the low
// bits of lo to f16's exponent range: absolute error <= 2^-39 x (row maximum) - far below what the row's large elements contribute to
// any dot product.  An fp32 product is the fp32-accumulated sum of THREE products, a_hi b_lo + a_lo b_hi + a_hi b_hi, each exact in the
// accumulator (11 x 11 significand bits); the dropped a_lo b_lo is <= 2^-22 |a b| (bf16x3 drops three terms of 2^-24 .. 2^-23 |a b|).
// The accumulator is un-scaled by 2^-(e_row + e_col) in the epilogue (exact).  Measured against float64 on the operands of the model
// and on adversarial ranges (tests/test_gpu_parity.py: test_gru_cell_f16x2_*; profiles/r06_h2_error_tables.txt): error at or below the
// vendor fp32 GEMM's and the bf16x3 kernel's - the accumulator rounds three times per 16-wide slice half instead of six.
// Non-finite operands: a row that holds Inf / NaN has a non-finite maximum, its scaled values are non-finite and every output that
// depends on it is NaN - the contract of the bf16x3 kernels ("Non-finite ... operands"); rows whose largest magnitude
// is below 2^-112 keep full range but lose relative precision (the scale exponent is clamped to 126).
//
// The row maxima of A are NOT computed here (a pass over 75 MB in front of a 120-us kernel): the kernel that produces c - the fused
// TarMAC message launch, csrc/tarmac_msg.hip, which reads every element of x and h anyway - writes max(|x_row|, |c_row|, |h_row|) per
// agent (`row_absmax`); a maximum is order-independent, so the result is deterministic.  Callers without that producer use
// gru_x3.hip.  The caller's maximum must bound the row (a smaller value overflows f16: the row's outputs become NaN, never a
// plausible wrong number).
//
// Kernel structure: gru_x3.hip's (512 threads = 128 agents x 64 hidden units, eight wavefronts of 32 x 32 x 4 accumulator sets on
// v_mfma_f32_32x32x16_f16, double-buffered LDS planes with one barrier per 32-wide K slice, staging interleaved with the first MFMA
// group) with two planes instead of three: 40 KB per LDS stage instead of 60, 8 fragment reads per slice half instead of 12, 3 VALU per
// staged element instead of 5.5 (v_pk_mul, v_cvt_pk_f16_f32, v_cvt_f32_f16, v_pk_fma, v_cvt_pk_f16_f32).
#include <type_traits>

#include "common.h"

namespace uavgnn {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 64, BK = 32, BJ = 64, NT = 256, ST = 68;
constexpr int NB = 2 * 3 * BJ * 4 / NT;              // weight chunks per thread and slice (6)
constexpr int RS = NT / 8;                           // rows per pass of the A loader (32)
constexpr int PA = BM * 4, PB = 3 * BJ * 4;            // 16-byte chunks per split plane of the A / B tile
constexpr int BUF = 2 * PA + 2 * PB;                   // chunks per buffer (40 KB)

__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ f16x8 as_frag(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// scale exponent of a row whose largest magnitude is `amax`: 2^se * amax lies in [2^14, 2^15) (se clamped to the normal range)
__device__ __forceinline__ int scale_exp(float amax) {
  const int e = static_cast<int>((__float_as_uint(amax) >> 23) & 0xffu);      // biased exponent; 255: Inf / NaN, 0: zero / subnormal
  return max(-126, min(126, 14 - (e - 127)));
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float(static_cast<unsigned>(e + 127) << 23); }

struct Split2 {
  unsigned hi, lo;   // two packed f16 each: low half = first element
};
// (x, y) already scaled -> hi + lo (round to nearest even both times; x - hi is exact in fp32)
__device__ __forceinline__ Split2 split_pair(float x, float y) {
  Split2 s;
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  s.hi = __builtin_bit_cast(unsigned, h);
  const f32x2 r = f32x2{x, y} - __builtin_convertvector(h, f32x2);
  s.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  return s;
}
// four consecutive k of one row (one 16-byte global load) scaled by `s` -> one 8-byte group per plane
__device__ __forceinline__ void stage4(unsigned short* p, int plane_stride, float4 v, float s) {
  const Split2 a = split_pair(v.x * s, v.y * s), b = split_pair(v.z * s, v.w * s);
  *reinterpret_cast<u32x2*>(p) = u32x2{a.hi, b.hi};
  *reinterpret_cast<u32x2*>(p + plane_stride) = u32x2{a.lo, b.lo};
}

// [W_ih | W_hh] -> f16 planes [2][3H][K_in] and [2][3H][H] + winv[3H] = 2^-e_row: one workgroup per output row (both matrices share
// the row's scale: the r and z accumulators sum both contractions)
__global__ __launch_bounds__(256) void split_planes_h2_kernel(const float* __restrict__ W_ih, int K_in, const float* __restrict__ W_hh,
                                                              int H, unsigned short* __restrict__ p_ih,
                                                              unsigned short* __restrict__ p_hh, float* __restrict__ winv) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* a = W_ih + static_cast<size_t>(row) * K_in;
  const float* b = W_hh + static_cast<size_t>(row) * H;
  float m = 0.f;
  for (int i = tid; i < K_in; i += 256) m = fmaxf(m, fabsf(a[i]));       // fmaxf drops NaN: a NaN weight is caught below
  for (int i = tid; i < H; i += 256) m = fmaxf(m, fabsf(b[i]));
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int se = scale_exp(m);
  const float s = pow2f(se);
  const size_t n_ih = static_cast<size_t>(3) * H * K_in, n_hh = static_cast<size_t>(3) * H * H;
  for (int i = 2 * tid; i < K_in; i += 512) {
    const Split2 sp = split_pair(a[i] * s, a[i + 1] * s);
    *reinterpret_cast<unsigned*>(p_ih + static_cast<size_t>(row) * K_in + i) = sp.hi;
    *reinterpret_cast<unsigned*>(p_ih + n_ih + static_cast<size_t>(row) * K_in + i) = sp.lo;
  }
  for (int i = 2 * tid; i < H; i += 512) {
    const Split2 sp = split_pair(b[i] * s, b[i + 1] * s);
    *reinterpret_cast<unsigned*>(p_hh + static_cast<size_t>(row) * H + i) = sp.hi;
    *reinterpret_cast<unsigned*>(p_hh + n_hh + static_cast<size_t>(row) * H + i) = sp.lo;
  }
  if (tid == 0) winv[row] = pow2f(-se);
}

// products of one fp32 product, smallest first: (a_hi b_lo) (a_lo b_hi) (a_hi b_hi)
template <bool SAVE>
__global__ __launch_bounds__(NT, 2) void gru_cell_fwd_h2_kernel(
    const float* __restrict__ inp, int ld_inp, int K1, const float* __restrict__ inp2, int ld_inp2, int K2,
    const float* __restrict__ h, int N, int H, const float* __restrict__ row_absmax,
    const unsigned short* __restrict__ Wih_p, const float* __restrict__ b_ih, const unsigned short* __restrict__ Whh_p,
    const float* __restrict__ b_hh, const float* __restrict__ winv, float* __restrict__ h_out, float* __restrict__ pre, int row_blocks) {
  __shared__ u32x4 smem[2 * BUF];   // buffer b: A planes [2][128][4] then B planes [2][192 = gate * 64 + unit][4]
  __shared__ float sInv[BM];        // 2^-e_row of the block's rows
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
  const int sa_w = lr * 32 + (((c4 >> 1) ^ swz32(lr)) * 8) + (c4 & 1) * 4;   // in f16 units inside an A plane
  // B loader: chunk q = tid + 512 i (i < 3, 1536 chunks): plane q / 768, row (q % 768) / 4 = gate * 64 + unit, chunk q % 4
  unsigned rowa[2], wrow[NB];
  float sca[2];
  int sbw[NB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    rowa[i] = static_cast<unsigned>(min(m0 + lr + RS * i, N - 1));
    sca[i] = pow2f(scale_exp(row_absmax[rowa[i]]));
  }
  if (tid < BM) sInv[tid] = pow2f(-scale_exp(row_absmax[min(m0 + tid, N - 1)]));
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int q = tid + NT * i, pl = q / PB, rem = q - pl * PB, row = rem >> 2, c = rem & 3;
    wrow[i] = static_cast<unsigned>(pl * 3 * H + (row >> 6) * H + j0 + (row & 63));
    sbw[i] = 2 * PA + pl * PB + row * 4 + (c ^ swz32(row));
  }
  const unsigned wc8 = 16u * (tid & 3);    // byte offset of the lane's 8-f16 chunk inside a weight slice
  const int n1 = K1 / BK, n12 = n1 + K2 / BK, ns = n12 + H / BK;   // slices of inp, of [inp || inp2], of everything
  // TWO register sets: the loads of slice t + 3 are issued while slice t computes (two iterations of latency cover; with one set -
  // one iteration - the staging waited for its loads: 26 of the kernel's 117 us in tools/h2_ablate.py)
  float4 ra[2][2];
  u32x4 rwb[NB];             // ONE set for the weight chunks (L2 hits: one iteration of cover), two for the activations
  unsigned oa[2], ow[NB];
  const char* __restrict__ Ab = reinterpret_cast<const char*>(inp);
  const char* __restrict__ Wb = reinterpret_cast<const char*>(Wih_p);
  int lt = 0, ltw = 0;                                       // the slice the next activation / weight load fetches
  auto set_a = [&](const float* base, int ld) {
    Ab = reinterpret_cast<const char*>(base);
#pragma unroll
    for (int i = 0; i < 2; ++i) oa[i] = 4u * (rowa[i] * static_cast<unsigned>(ld) + 4u * c4);
  };
  auto set_w = [&](const unsigned short* base, int K) {
    Wb = reinterpret_cast<const char*>(base);
#pragma unroll
    for (int i = 0; i < NB; ++i) ow[i] = 2u * wrow[i] * static_cast<unsigned>(K) + wc8;
  };
  set_a(inp, ld_inp);
  set_w(Wih_p, K1 + K2);
  auto gload_a = [&](auto set) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[S][i] = *reinterpret_cast<const float4*>(Ab + oa[i]);
  };
  auto advance_w = [&]() {
    if (ltw + 1 >= ns) return;
    ++ltw;
    Wb += 2 * BK;
    if (ltw == n12) set_w(Whh_p, H);
  };
  auto gload_w = [&](auto) {
#pragma unroll
    for (int i = 0; i < NB; ++i) rwb[i] = *reinterpret_cast<const u32x4*>(Wb + ow[i]);
    advance_w();
  };
  auto advance = [&]() {      // after the activation loads of slice lt were issued; past the last slice the cursor stays on it
    if (lt + 1 >= ns) return;
    ++lt;
    Ab += 4 * BK;
    if (lt == n1 && n12 > n1) set_a(inp2, ld_inp2);
    if (lt == n12) set_a(h, H);
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  auto gload = [&](auto set) {
    gload_a(set);
    advance();
  };
  auto lstore_b = [&](int buf, auto) {
    u32x4* sb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < NB; ++i) sb[sbw[i]] = rwb[i];
  };
  auto lstore_a = [&](int buf, int i, auto set) {
    constexpr int S = decltype(set)::value;
    unsigned short* sa = reinterpret_cast<unsigned short*>(smem + buf * BUF) + sa_w;
    stage4(sa + RS * i * 32, PA * 8, ra[S][i], sca[i]);
  };
  auto lstore = [&](int buf, auto set) {
    lstore_a(buf, 0, set);
    lstore_a(buf, 1, set);
    lstore_b(buf, set);
  };
#ifdef UAVGNN_H2_DBG
#define UAVGNN_H2_DBG_ UAVGNN_H2_DBG
#else
#define UAVGNN_H2_DBG_ 0
#endif
  struct Half {
    f16x8 a[2], b[3][2];   // [plane], [gate][plane]
  };
#define UAVGNN_H2_READ(F, buf, kh)                                                                                 \
  if (!(UAVGNN_H2_DBG_ & 16) || t == 0) {                                                                          \
    const u32x4* sb = smem + (buf) * BUF;                                                                          \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) F.a[pl] = as_frag(sb[pl * PA + (wm + l32) * 4 + ((2 * (kh) + lh) ^ sw)]); \
    _Pragma("unroll") for (int gate = 0; gate < 3; ++gate) _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)        \
        F.b[gate][pl] = as_frag(sb[2 * PA + pl * PB + (gate * BJ + wc + l32) * 4 + ((2 * (kh) + lh) ^ sw)]);       \
  }
#define UAVGNN_H2_TERM(ia, ib)                       \
  if (!(UAVGNN_H2_DBG_ & 8)) {                       \
    acc[0] = mfma32(F.a[ia], F.b[0][ib], acc[0]);    \
    acc[1] = mfma32(F.a[ia], F.b[1][ib], acc[1]);    \
    acc[NSET] = mfma32(F.a[ia], F.b[2][ib], acc[NSET]); \
  }
#define UAVGNN_H2_MFMA(F_, NSET_)                                  \
  {                                                                \
    constexpr int NSET = NSET_;                                    \
    const Half& F = F_;                                            \
    UAVGNN_H2_TERM(0, 1) UAVGNN_H2_TERM(1, 0) UAVGNN_H2_TERM(0, 0) \
  }

  gload(Set0{});          // slice 0
  gload_w(Set0{});
  lstore(0, Set0{});
  gload(Set1{});          // slice 1
  gload(Set0{});          // slice 2
  gload_w(Set0{});        // weights of slice 1
  __syncthreads();
  // Software pipeline as in gru_x3.hip: the fragment reads of a half are issued one MFMA group (9 MFMAs) before their use; iteration
  // t stages slice t + 1 into the other buffer INSIDE its first MFMA group (an independent VALU / LDS / memory instruction issues in
  // the shadow of an executing MFMA only when it follows it in the instruction stream) and starts the loads of slice t + 2.
  Half f0, f1;
  int t = 0;
  UAVGNN_H2_READ(f0, 0, 0)
#ifndef UAVGNN_H2_DBG
#define UAVGNN_H2_DBG 0   /* timing experiments (tools/h2_ablate.py; results are WRONG): bit 0 no global loads in the loop, 1 no staging, 2 no slice loop at all, 3 no MFMAs, 4 no fragment reads in the loop */
#endif
#define UAVGNN_H2_STEP(NSET_, SET_)                        \
  {                                                        \
    constexpr int NSET = NSET_;                            \
    UAVGNN_H2_READ(f1, t & 1, 1)                           \
    __builtin_amdgcn_sched_barrier(0);                     \
    const Half& F = f0;                                    \
    if (!(UAVGNN_H2_DBG & 2)) lstore_b((t + 1) & 1, SET_{}); \
    if (!(UAVGNN_H2_DBG & 1)) gload_w(SET_{});             \
    UAVGNN_H2_TERM(0, 1)                                   \
    _Pragma("unroll") for (int sg = 0; sg < 3; ++sg) {     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   \
    }                                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    if (!(UAVGNN_H2_DBG & 2)) { lstore_a((t + 1) & 1, 0, SET_{}); lstore_a((t + 1) & 1, 1, SET_{}); } \
    UAVGNN_H2_TERM(1, 0)                                   \
    _Pragma("unroll") for (int sg = 0; sg < 3; ++sg) {     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   \
    }                                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    if (!(UAVGNN_H2_DBG & 1)) gload_a(SET_{});             \
    UAVGNN_H2_TERM(0, 0)                                   \
    _Pragma("unroll") for (int sg = 0; sg < 3; ++sg) {     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   \
    }                                                      \
    __builtin_amdgcn_sched_barrier(0);                     \
    if (!(UAVGNN_H2_DBG & 1)) advance();                   \
  }                                                        \
  __syncthreads();                                         \
  UAVGNN_H2_READ(f0, (t + 1) & 1, 0)                       \
  __builtin_amdgcn_sched_barrier(0);                       \
  UAVGNN_H2_MFMA(f1, NSET_)                                \
  __builtin_amdgcn_sched_barrier(0);
  if (UAVGNN_H2_DBG & 4) t = ns;
  // iteration t stages slice t + 1 out of register set (t + 1) & 1 and refills that set with slice t + 3
  for (; t + 1 < n12; ++t) {
    UAVGNN_H2_STEP(2, Set1)
    ++t;
    UAVGNN_H2_STEP(2, Set0)
  }
  if (t < n12) {            // an odd number of input slices: the pairs of the second loop start on an odd slice
    UAVGNN_H2_STEP(2, Set1)
    ++t;
    UAVGNN_H2_STEP(3, Set0)
    ++t;
  }
  for (; t + 1 < ns; ++t) {
    UAVGNN_H2_STEP(3, Set1)
    ++t;
    UAVGNN_H2_STEP(3, Set0)
  }
  if (t < ns) {             // (odd slice count)
    UAVGNN_H2_STEP(3, Set1)
    ++t;
  }
#undef UAVGNN_H2_STEP
#undef UAVGNN_H2_MFMA
#undef UAVGNN_H2_READ
#undef UAVGNN_H2_TERM
  __syncthreads();   // the last iteration's read of the stale buffer must not race the epilogue's tile
  // ---- epilogue on the D layout: lane l holds column l % 32, register i holds row 8 (i / 4) + 4 (l / 32) + i % 4 ----------------
  float* sH = reinterpret_cast<float*>(smem);             // [128][ST] fp32 tile: h in, h' out, 16-byte row-contiguous HBM accesses
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + NT * q, row = idx >> 4, cc = idx & 15;
    *reinterpret_cast<float4*>(sH + row * ST + 4 * cc) =
        *reinterpret_cast<const float4*>(h + static_cast<size_t>(min(m0 + row, N - 1)) * H + j0 + 4 * cc);
  }
  __syncthreads();
  const int c = j0 + wc + l32;
  const float b_r = b_ih[c] + b_hh[c], b_z = b_ih[H + c] + b_hh[H + c], b_in = b_ih[2 * H + c], b_hn = b_hh[2 * H + c];
  const float ci_r = winv[c], ci_z = winv[H + c], ci_n = winv[2 * H + c];   // 2^-e of the three weight rows of this hidden unit
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int lrow = wm + 8 * (i >> 2) + 4 * lh + (i & 3);
    const int row = m0 + lrow;
    const float ri = sInv[lrow];
    // two exact power-of-two factors (their product alone may leave the fp32 range), then the bias: one rounding, as before
    const float pr = acc[0][i] * ri * ci_r + b_r, pz = acc[1][i] * ri * ci_z + b_z;
    const float gin = acc[2][i] * ri * ci_n + b_in, ghn = acc[3][i] * ri * ci_n + b_hn;
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

extern "C" int uavgnn_gru_cell_h2_supported(int K_in, int H) {
  return (K_in >= BK && K_in % BK == 0 && H >= BJ && H % BJ == 0) ? 1 : 0;
}

// f16 planes [2][3H][K_in] + [2][3H][H], then winv[3H] floats
extern "C" long long uavgnn_gru_cell_h2_workspace_bytes(int K_in, int H) {
  if (K_in <= 0 || H <= 0) return 0;
  return 2LL * 3 * H * (static_cast<long long>(K_in) + H) * 2 + 4LL * 3 * H;
}

extern "C" int uavgnn_gru_split_weights_h2(const float* W_ih, int K_in, const float* W_hh, int H, void* planes,
                                           uavgnn_stream_t stream) {
  if (!W_ih || !W_hh || !planes || K_in <= 0 || H <= 0) return UAVGNN_EINVAL;
  if ((K_in & 1) || (H & 1) || ((reinterpret_cast<uintptr_t>(W_ih) | reinterpret_cast<uintptr_t>(W_hh)) & 7) ||
      (reinterpret_cast<uintptr_t>(planes) & 15))
    return UAVGNN_EUNSUPPORTED;
  unsigned short* p0 = static_cast<unsigned short*>(planes);
  unsigned short* p1 = p0 + 6LL * H * K_in;
  float* winv = reinterpret_cast<float*>(p1 + 6LL * H * H);
  hipLaunchKernelGGL(split_planes_h2_kernel, dim3(3 * H), dim3(256), 0, static_cast<hipStream_t>(stream), W_ih, K_in, W_hh, H, p0, p1,
                     winv);
  return launch_status();
}

// The GRU cell of uavgnn_gru_cell_fwd_x3_cat on the f16x2 arithmetic.  row_absmax [N]: an upper bound of max |.| over the row of
// [inp || inp2 || h] for every agent, tight to within its power of two (uavgnn_tarmac_msg_fwd_rowmax writes it); planes:
// uavgnn_gru_split_weights_h2.
extern "C" int uavgnn_gru_cell_fwd_h2(const float* inp, int ld_inp, int K1, const float* inp2, int ld_inp2, int K2, const float* h,
                                      int N, int H, const float* row_absmax, const void* planes, const float* b_ih,
                                      const float* b_hh, float* h_out, float* pre_save, uavgnn_stream_t stream) {
  const int K_in = K1 + K2;
  if (N < 0 || !inp || !h || !planes || !b_ih || !b_hh || !h_out || !row_absmax || ld_inp < K1 || K2 < 0 ||
      (K2 > 0 && (!inp2 || ld_inp2 < K2)))
    return UAVGNN_EINVAL;
  if (K2 == 0) {
    inp2 = inp;
    ld_inp2 = ld_inp;
  }
  if (!uavgnn_gru_cell_h2_supported(K_in, H) || K1 < BK || (K1 % BK) || (K2 % BK) || (ld_inp & 3) || (ld_inp2 & 3) ||
      ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(inp2) | reinterpret_cast<uintptr_t>(h) |
        reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(h_out)) & 15))
    return UAVGNN_EUNSUPPORTED;
  // the kernel addresses its operands by 32-bit BYTE offsets from the base pointers (global_load with an SGPR base)
  const long long ld_max = ld_inp > ld_inp2 ? (ld_inp > H ? ld_inp : H) : (ld_inp2 > H ? ld_inp2 : H);
  if (4LL * N * ld_max >= (1LL << 32) || 12LL * H * (K_in > H ? K_in : H) >= (1LL << 32)) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  const unsigned short* p0 = static_cast<const unsigned short*>(planes);
  const unsigned short* p1 = p0 + 6LL * H * K_in;
  const float* winv = reinterpret_cast<const float*>(p1 + 6LL * H * H);
  const int row_blocks = (N + BM - 1) / BM, rb8 = ((row_blocks + 7) / 8) * 8;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(rb8 * (H / BJ)), block(NT);
  if (pre_save != nullptr)
    hipLaunchKernelGGL(gru_cell_fwd_h2_kernel<true>, grid, block, 0, st, inp, ld_inp, K1, inp2, ld_inp2, K2, h, N, H, row_absmax, p0, b_ih,
                       p1, b_hh, winv, h_out, pre_save, row_blocks);
  else
    hipLaunchKernelGGL(gru_cell_fwd_h2_kernel<false>, grid, block, 0, st, inp, ld_inp, K1, inp2, ld_inp2, K2, h, N, H, row_absmax, p0, b_ih,
                       p1, b_hh, winv, h_out, pre_save, row_blocks);
  return launch_status();
}

// max |.| per row of up to three row-major pieces (a2 / a3 may be NULL): the `row_absmax` of uavgnn_gru_cell_fwd_h2 for callers without
// a producer that writes it (tests, probes: a pass over the operand the shipped path does not pay).  One wavefront per row.
namespace uavgnn {
namespace {
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ a1, int ld1, int K1, const float* __restrict__ a2,
                                                         int ld2, int K2, const float* __restrict__ a3, int ld3, int K3, int N,
                                                         float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  float m = 0.f;
  bool bad = false;
  auto piece = [&](const float* a, int ld, int K) {
    if (a == nullptr) return;
    const float* r = a + static_cast<size_t>(row) * ld;
    for (int i = lane; i < K; i += 64) {
      const float v = fabsf(r[i]);
      bad |= !(v <= 3.4028234663852886e38f);     // Inf / NaN
      m = fmaxf(m, v);
    }
  };
  piece(a1, ld1, K1);
  piece(a2, ld2, K2);
  piece(a3, ld3, K3);
  m = wave_max(m);
  if (__any(bad)) m = INFINITY;
  if (lane == 0) out[row] = m;
}
}  // namespace
}  // namespace uavgnn

extern "C" int uavgnn_row_absmax(const float* a1, int ld1, int K1, const float* a2, int ld2, int K2, const float* a3, int ld3, int K3,
                                 int N, float* out, uavgnn_stream_t stream) {
  if (N < 0 || !a1 || !out || K1 <= 0 || ld1 < K1 || (a2 && (K2 <= 0 || ld2 < K2)) || (a3 && (K3 <= 0 || ld3 < K3))) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(row_absmax_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a1, ld1, K1, a2, ld2, K2, a3,
                     ld3, K3, N, out);
  return launch_status();
}
