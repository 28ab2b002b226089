// Hazard probe (gfx950): how many wait states does a VALU read of an MFMA result need, and does the answer differ
// between a plain v_add_f32 and a packed v_pk_add_f32 (which runs its two halves in two passes)?
// Why: the matrix-core K1 backward (gatv2.hip, UAVGNN_K1_BWD_MFMA=1) is bit-reproducible when the TU is compiled
// with -target-feature -packed-fp32-ops and is NOT when the compiler forms v_pk_* instructions; the wrong values sit
// in lanes 48-63 of the LOW halves of register pairs.  This program issues
//     v_mfma_f32_16x16x32_bf16 D, A, B, 0 ; [FILL x v_mfma to other registers] ; s_nop (NOPS-1) ; consumer(D)
// from hand-placed registers, for NOPS = 0..15, and counts per (lane group, element) how often the consumer saw
// something else than the MFMA result (the destination registers hold a sentinel before the MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_pk_hazard.hip -o tools/ubench/bin/mfma_pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: two v_pk_add_f32; 1: four v_add_f32; 2: two v_pk_mul_f32; 3: two v_pk_fma_f32 (D as the multiplicand)
// FILL: independent MFMAs between the producer and the consumer (the compiler counts each as ONE wait state)
template <int NOPS, int MODE, int FILL>
__global__ __launch_bounds__(512) void probe(const s16x8* a_in, const s16x8* b_in, const f32x4* want, unsigned* bad,
                                              int iters, int contend) {
  const int lane = threadIdx.x & 63;
  const s16x8 a = a_in[lane], b = b_in[lane];
  if (contend && (threadIdx.x >> 6) < 4) {   // partner waves (one per SIMD): MFMAs back to back for the whole run
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int it = 0; it < iters * 6; ++it)
      for (int m = 0; m < 4; ++m)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if (s == 123.456f) bad[15] = 1u;
    return;
  }
  const f32x4 w = want[lane];
  unsigned miss[4] = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    f32x4 got;
    float zero = 0.f;
    f32x2 zero2 = {0.f, 0.f}, one2 = {1.f, 1.f};
    asm volatile(
        "v_mov_b32 v100, 0xc4424000\n v_mov_b32 v101, 0xc4424000\n v_mov_b32 v102, 0xc4424000\n v_mov_b32 v103, 0xc4424000\n"
        "s_nop 7\n"
        "v_mfma_f32_16x16x32_bf16 v[100:103], %[a], %[b], 0\n"
        ".rept %c[fill]\n v_mfma_f32_16x16x32_bf16 v[108:111], %[a], %[b], 0\n .endr\n"
        ".rept %c[nops]\n s_nop 0\n .endr\n"
        ".if %c[mode] == 0\n"
        " v_pk_add_f32 v[104:105], v[100:101], %[z2]\n v_pk_add_f32 v[106:107], v[102:103], %[z2]\n"
        ".elseif %c[mode] == 1\n"
        " v_add_f32 v104, v100, %[z]\n v_add_f32 v105, v101, %[z]\n v_add_f32 v106, v102, %[z]\n v_add_f32 v107, v103, %[z]\n"
        ".elseif %c[mode] == 2\n"
        " v_pk_mul_f32 v[104:105], v[100:101], %[o2]\n v_pk_mul_f32 v[106:107], v[102:103], %[o2]\n"
        ".else\n"
        " v_pk_fma_f32 v[104:105], v[100:101], %[o2], %[z2]\n v_pk_fma_f32 v[106:107], v[102:103], %[o2], %[z2]\n"
        ".endif\n"
        "s_nop 15\n s_nop 15\n"
        "v_mov_b32 %[g0], v104\n v_mov_b32 %[g1], v105\n v_mov_b32 %[g2], v106\n v_mov_b32 %[g3], v107\n"
        : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3])
        : [a] "v"(a), [b] "v"(b), [z] "v"(zero), [z2] "v"(zero2), [o2] "v"(one2), [nops] "n"(NOPS), [mode] "n"(MODE), [fill] "n"(FILL)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    for (int r = 0; r < 4; ++r) miss[r] += (got[r] != w[r]) ? 1u : 0u;
  }
  for (int r = 0; r < 4; ++r)
    if (miss[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], miss[r]);
}


// Cross-wave probe: 8 waves per workgroup (two per SIMD).  Waves 0-3 issue MFMAs back to back (PARTNER 1: bf16 16x16x32,
// 2: fp32 16x16x4, 3: a chain of plain v_fma_f32 instead, 0: nothing); waves 4-7 run a chain of fp32 FMAs on small integers
// (exact) and compare with the closed form.  FORM selects the instruction of the chain:
//   0 v_fma_f32 (two per step)      1 v_pk_fma_f32 (no operand select)     2 v_pk_fma_f32 op_sel:[0,1,0]
//   3 v_pk_fma_f32 op_sel_hi:[1,0,1]  4 v_pk_mul_f32 op_sel:[0,1]  + v_pk_add   5 v_pk_add_f32 op_sel_hi:[0,1]
template <int FORM, int PARTNER>
__global__ __launch_bounds__(512) void cross(const s16x8* a_in, const s16x8* b_in, unsigned* bad, float* sink, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (PARTNER == 0 || PARTNER == 4) return;
    const s16x8 a = a_in[lane], b = b_in[lane];
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    float fa = 1.f, fb = 0.5f;
    for (int it = 0; it < iters * 8; ++it)
      for (int m = 0; m < 4; ++m) {
        if (PARTNER == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
        if (PARTNER == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fa), "v"(fb));
        if (PARTNER == 3) for (int r = 0; r < 4; ++r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[m][r]) : "v"(fa), "v"(fb));
      }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if (s == 123.456f) sink[0] = s;
    return;
  }
  unsigned miss[2] = {0u, 0u};
  const f32x2 one2 = {1.f, 1.f}, inc2 = {1.f, 2.f}, sel2 = {0.f, 1.f}, les2 = {1.f, 0.f};
  const float one = 1.f;
  const s16x8 oa = a_in[lane], ob = b_in[lane];
  f32x4 own = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x2 x = {static_cast<float>(lane), static_cast<float>(2 * lane)};
    for (int k = 0; k < 64; ++k) {
      if (FORM == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(one), "v"(inc2[0]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(one), "v"(inc2[1]));
      }
      if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(one2), "v"(inc2));
      if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(x) : "v"(sel2), "v"(inc2));
      if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(les2), "v"(inc2));
      if (FORM == 4) {
        asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(x) : "v"(sel2));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(inc2));
      }
      if (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[1,0,0]" : "+v"(x) : "v"(sel2), "v"(inc2));
      if (FORM == 7) {   // lo += s2.hi (= 1), hi += s2.hi (= 1); then hi += 1
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1]" : "+v"(x) : "v"(one2), "v"(sel2));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(sel2));
      }
      if (FORM == 8) {   // y.lo = x.hi by v_pk_mov_b32 with op_sel; x itself runs the plain packed chain
        f32x2 y;
        asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(y) : "v"(x));
        miss[0] += (y[0] != x[1]) ? 1u : 0u;
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(one2), "v"(inc2));
      }
      if (PARTNER == 4 && (k & 3) == 0)   // the chain wave's OWN MFMA in flight (no partner waves)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(own) : "v"(oa), "v"(ob));
      if (FORM == 5) {   // x += (inc.lo, inc.lo) then x.hi += 1
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(x) : "v"(inc2));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(sel2));
      }
    }
    miss[0] += (x[0] != static_cast<float>(lane + 64)) ? 1u : 0u;
    miss[1] += (x[1] != static_cast<float>(2 * lane + 128)) ? 1u : 0u;
  }
  if (own[0] + own[1] + own[2] + own[3] == 123.456f) sink[1] = own[0];
  for (int r = 0; r < 2; ++r)
    if (miss[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], miss[r]);
}

template <int FORM, int PARTNER>
static void run_cross(const s16x8* a, const s16x8* b, unsigned* bad, float* sink, int blocks) {
  static const char* forms[] = {"v_fma_f32 x2", "v_pk_fma_f32", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
                                "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel:[1,0,0]",
                                "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mov_b32 op_sel:[1,0]"};
  static const char* partners[] = {"none", "mfma bf16 16x16x32", "mfma f32 16x16x4", "v_fma_f32 chain", "none, own bf16 mfma"};
  HIP_OK(hipMemset(bad, 0, 16 * sizeof(unsigned)));
  cross<FORM, PARTNER><<<blocks, 512>>>(a, b, bad, sink, 2000);
  HIP_OK(hipDeviceSynchronize());
  unsigned h[16];
  HIP_OK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("cross-wave %-32s partner waves: %-20s %5d workgroups: misses by [lanes 0-15|16-31|32-47|48-63] x [lo hi]:", forms[FORM],
         partners[PARTNER], blocks);
  for (int g = 0; g < 4; ++g) printf(" | %u %u", h[g * 4], h[g * 4 + 1]);
  printf("\n");
}

template <int FORM>
static void sweep_cross(const s16x8* a, const s16x8* b, unsigned* bad, float* sink) {
  for (int blocks : {256, 512, 1024, 2048}) {
    run_cross<FORM, 1>(a, b, bad, sink, blocks);
    run_cross<FORM, 2>(a, b, bad, sink, blocks);
    run_cross<FORM, 3>(a, b, bad, sink, blocks);
    run_cross<FORM, 0>(a, b, bad, sink, blocks);
    run_cross<FORM, 4>(a, b, bad, sink, blocks);
  }
}

// Same-wave distance: v_mfma (independent registers) ; NOPS x s_nop 0 ; v_pk_fma_f32 op_sel:[0,1,0] TWICE, the second reading
// the first one's result (a single packed FMA after the MFMA never failed; the failing shape is a dependent pair).
// One wave per SIMD (256 workgroups of 256 threads), nothing else on the chip.
template <int NOPS, int KIND>
__global__ __launch_bounds__(256) void own_gap(const s16x8* a_in, const s16x8* b_in, unsigned* bad, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  const s16x8 oa = a_in[lane], ob = b_in[lane];
  f32x4 own = {0.f, 0.f, 0.f, 0.f};
  f32x16 own16;
  for (int i = 0; i < 16; ++i) own16[i] = 0.f;
  const f32x2 inc2 = {1.f, 2.f}, sel2 = {0.f, 1.f};
  const float fa = 1.f, fb = 0.5f;
  unsigned miss[2] = {0u, 0u};
  for (int it = 0; it < iters; ++it) {
    f32x2 x = {static_cast<float>(lane), static_cast<float>(2 * lane)};
    for (int k = 0; k < 64; ++k) {
      if (KIND == 0)
        asm volatile("v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .rept %c6\n s_nop 0\n .endr\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]"
                     : "+v"(x), "+v"(own) : "v"(oa), "v"(ob), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 1)   // the fp32 MFMA (one VGPR per operand)
        asm volatile("v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n .rept %c6\n s_nop 0\n .endr\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]"
                     : "+v"(x), "+v"(own) : "v"(fa), "v"(fb), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 3)   // the MFMA BETWEEN the two dependent packed FMAs, every step
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .rept %c6\n s_nop 0\n .endr\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]"
                     : "+v"(x), "+v"(own) : "v"(oa), "v"(ob), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 5)   // the wait states between the FIRST packed FMA and the MFMA that follows it
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n .rept %c6\n s_nop 0\n .endr\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n s_nop 7\n s_nop 7\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n s_nop 7"
                     : "+v"(x), "+v"(own) : "v"(oa), "v"(ob), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 6)   // the 32x32x16 bf16 MFMA (the GRU cell's / the GEMMs') right behind the packed FMA
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n .rept %c6\n s_nop 0\n .endr\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n s_nop 7\n s_nop 7\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n s_nop 7"
                     : "+v"(x), "+v"(own16) : "v"(oa), "v"(ob), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 7)   // fp8 16x16x128 is not used here; the f16 16x16x32 MFMA
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n .rept %c6\n s_nop 0\n .endr\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n s_nop 7\n s_nop 7\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n s_nop 7"
                     : "+v"(x), "+v"(own) : "v"(oa), "v"(ob), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 8)   // the fp32 16x16x4 MFMA right behind the packed FMA (control)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n .rept %c6\n s_nop 0\n .endr\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n s_nop 7\n s_nop 7\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n s_nop 7"
                     : "+v"(x), "+v"(own) : "v"(fa), "v"(fb), "v"(sel2), "v"(inc2), "n"(NOPS));
      if (KIND == 4) {  // ... on every fourth step only (three plain steps of two packed FMAs in between)
        if ((k & 3) == 0)
          asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .rept %c6\n s_nop 0\n .endr\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]"
                       : "+v"(x), "+v"(own) : "v"(oa), "v"(ob), "v"(sel2), "v"(inc2), "n"(NOPS));
        else
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(x) : "v"(sel2), "v"(inc2));
      }
      if (KIND == 2)   // the older bf16 MFMA with 64-bit operands
        asm volatile("v_mfma_f32_16x16x16_bf16 %1, %2, %3, %1\n .rept %c6\n s_nop 0\n .endr\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0]"
                     : "+v"(x), "+v"(own) : "v"(inc2), "v"(sel2), "v"(sel2), "v"(inc2), "n"(NOPS));
    }
    miss[0] += (x[0] != static_cast<float>(lane + 128)) ? 1u : 0u;
    miss[1] += (x[1] != static_cast<float>(2 * lane + 256)) ? 1u : 0u;
  }
  if (own[0] + own[1] + own[2] + own[3] + own16[0] + own16[7] == 123.456f) sink[1] = own[0];
  for (int r = 0; r < 2; ++r)
    if (miss[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], miss[r]);
}

template <int NOPS, int KIND>
static void run_gap(const s16x8* a, const s16x8* b, unsigned* bad, float* sink) {
  static const char* kinds[] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_bf16",
                                "pk ; v_mfma_f32_16x16x32_bf16 (every step)", "pk ; v_mfma_f32_16x16x32_bf16 (every 4th step)",
                                "pk ; N x s_nop 0 ; v_mfma_f32_16x16x32_bf16 ; 16 wait states",
                                "pk ; N x s_nop 0 ; v_mfma_f32_32x32x16_bf16 ; 16 wait states", "pk ; N x s_nop 0 ; v_mfma_f32_16x16x32_f16 ; 16 wait states",
                                "pk ; N x s_nop 0 ; v_mfma_f32_16x16x4_f32 ; 16 wait states"};
  HIP_OK(hipMemset(bad, 0, 16 * sizeof(unsigned)));
  own_gap<NOPS, KIND><<<256, 256>>>(a, b, bad, sink, 500);
  HIP_OK(hipDeviceSynchronize());
  unsigned h[16];
  HIP_OK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("same wave: %s ; %2d x s_nop 0 ; %s v_pk_fma_f32 op_sel:[0,1,0]: misses by [lane group] x [lo hi]:", kinds[KIND], NOPS, KIND >= 5 ? "(N counts the nops in FRONT of the MFMA) the dependent" : KIND >= 3 ? "the dependent" : "2 dependent");
  for (int g = 0; g < 4; ++g) printf(" | %u %u", h[g * 4], h[g * 4 + 1]);
  printf("\n");
}

template <int KIND>
static void sweep_gap(const s16x8* a, const s16x8* b, unsigned* bad, float* sink) {
  run_gap<0, KIND>(a, b, bad, sink); run_gap<1, KIND>(a, b, bad, sink); run_gap<2, KIND>(a, b, bad, sink);
  run_gap<3, KIND>(a, b, bad, sink); run_gap<4, KIND>(a, b, bad, sink); run_gap<5, KIND>(a, b, bad, sink);
  run_gap<6, KIND>(a, b, bad, sink); run_gap<7, KIND>(a, b, bad, sink); run_gap<8, KIND>(a, b, bad, sink);
  run_gap<10, KIND>(a, b, bad, sink); run_gap<12, KIND>(a, b, bad, sink); run_gap<16, KIND>(a, b, bad, sink);
}

__device__ inline unsigned pack2(short lo, short hi) { return static_cast<unsigned short>(lo) | (static_cast<unsigned>(static_cast<unsigned short>(hi)) << 16); }

// WAR probe: v_mfma D, A, B, C ; NOPS x s_nop 0 ; VALU overwrites one source (WHICH 0: A, 1: B, 2: C = D's old value).
// PRE independent MFMAs are issued first so that the tested one may have to queue.  With `contend` the four oldest
// waves of the 8-wave workgroup keep the matrix pipe of every SIMD busy.
template <int NOPS, int WHICH, int PRE>
__global__ __launch_bounds__(512) void war(const s16x8* a_in, const s16x8* b_in, const f32x4* want, unsigned* bad, int iters,
                                            int contend) {
  const int lane = threadIdx.x & 63;
  const s16x8 a = a_in[lane], b = b_in[lane];
  if (contend && (threadIdx.x >> 6) < 4) {
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int it = 0; it < iters * 6; ++it)
      for (int m = 0; m < 4; ++m)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if (s == 123.456f) bad[15] = 1u;
    return;
  }
  const f32x4 w = want[lane];
  unsigned miss[4] = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    f32x4 got;
    asm volatile(
        "v_mov_b32 v100, %[a0]\n v_mov_b32 v101, %[a1]\n v_mov_b32 v102, %[a2]\n v_mov_b32 v103, %[a3]\n"   // A copy
        "v_mov_b32 v104, %[b0]\n v_mov_b32 v105, %[b1]\n v_mov_b32 v106, %[b2]\n v_mov_b32 v107, %[b3]\n"   // B copy
        "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"                   // C = 0
        "s_nop 7\n"
        ".rept %c[pre]\n v_mfma_f32_16x16x32_bf16 v[108:111], %[a], %[b], 0\n .endr\n"
        "v_mfma_f32_16x16x32_bf16 v[116:119], v[100:103], v[104:107], v[112:115]\n"
        ".rept %c[nops]\n s_nop 0\n .endr\n"
        ".if %c[which] == 0\n"
        " v_pk_mov_b32 v[100:101], v[108:109], v[108:109]\n v_pk_mov_b32 v[102:103], v[108:109], v[108:109]\n"
        ".elseif %c[which] == 1\n"
        " v_pk_mov_b32 v[104:105], v[100:101], v[100:101]\n v_pk_mov_b32 v[106:107], v[100:101], v[100:101]\n"
        ".else\n"
        " v_pk_mov_b32 v[112:113], v[100:101], v[100:101]\n v_pk_mov_b32 v[114:115], v[100:101], v[100:101]\n"
        ".endif\n"
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_mov_b32 %[g0], v116\n v_mov_b32 %[g1], v117\n v_mov_b32 %[g2], v118\n v_mov_b32 %[g3], v119\n"
        : [g0] "=&v"(got[0]), [g1] "=&v"(got[1]), [g2] "=&v"(got[2]), [g3] "=&v"(got[3])
        : [a] "v"(a), [b] "v"(b), [a0] "v"(pack2(a[0], a[1])), [a1] "v"(pack2(a[2], a[3])), [a2] "v"(pack2(a[4], a[5])),
          [a3] "v"(pack2(a[6], a[7])), [b0] "v"(pack2(b[0], b[1])), [b1] "v"(pack2(b[2], b[3])),
          [b2] "v"(pack2(b[4], b[5])), [b3] "v"(pack2(b[6], b[7])), [nops] "n"(NOPS), [which] "n"(WHICH), [pre] "n"(PRE)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
          "v114", "v115", "v116", "v117", "v118", "v119");
    for (int r = 0; r < 4; ++r) miss[r] += (got[r] != w[r]) ? 1u : 0u;
  }
  for (int r = 0; r < 4; ++r)
    if (miss[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], miss[r]);
}

template <int NOPS, int WHICH, int PRE>
static void run_war(const s16x8* a, const s16x8* b, const f32x4* want, unsigned* bad, int blocks, int contend) {
  HIP_OK(hipMemset(bad, 0, 16 * sizeof(unsigned)));
  war<NOPS, WHICH, PRE><<<blocks, contend ? 512 : 256>>>(a, b, want, bad, 2000, contend);
  HIP_OK(hipDeviceSynchronize());
  unsigned h[16];
  HIP_OK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long total = 0;
  for (int i = 0; i < 15; ++i) total += h[i];
  printf("WAR on %s, %d MFMAs queued before, nops %d, blocks %4d%s: misses %lu\n", WHICH == 0 ? "A" : WHICH == 1 ? "B" : "C", PRE, NOPS,
         blocks, contend ? " +partner" : "", total);
}

template <int WHICH, int PRE>
static void sweep_war(const s16x8* a, const s16x8* b, const f32x4* want, unsigned* bad) {
  for (int contend : {0, 1})
    for (int blocks : {1, 1024}) {
      run_war<0, WHICH, PRE>(a, b, want, bad, blocks, contend);
      run_war<1, WHICH, PRE>(a, b, want, bad, blocks, contend);
      run_war<2, WHICH, PRE>(a, b, want, bad, blocks, contend);
      run_war<4, WHICH, PRE>(a, b, want, bad, blocks, contend);
    }
}

static unsigned short bf16(float f) { unsigned u; memcpy(&u, &f, 4); return static_cast<unsigned short>(u >> 16); }

template <int NOPS, int MODE, int FILL>
static void run_one(const s16x8* a, const s16x8* b, const f32x4* want, unsigned* bad, int blocks) {
  const int contend = blocks < 0;
  blocks = contend ? -blocks : blocks;
  HIP_OK(hipMemset(bad, 0, 16 * sizeof(unsigned)));
  probe<NOPS, MODE, FILL><<<blocks, contend ? 512 : 256>>>(a, b, want, bad, 2000, contend);
  HIP_OK(hipDeviceSynchronize());
  unsigned h[16];
  HIP_OK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long total = 0;
  for (unsigned v : h) total += v;
  printf("mode %d fill %d nops %2d blocks %5d%s: misses %10lu", MODE, FILL, NOPS, blocks, contend ? " +dense-MFMA partner wave per SIMD" : "", total);
  if (total) {
    printf("   [lanes 0-15|16-31|32-47|48-63] x [elem 0..3]:");
    for (int g = 0; g < 4; ++g) { printf(" |"); for (int r = 0; r < 4; ++r) printf(" %u", h[g * 4 + r]); }
  }
  printf("\n");
}

template <int MODE, int FILL>
static void sweep(const s16x8* a, const s16x8* b, const f32x4* want, unsigned* bad, int blocks) {
  run_one<0, MODE, FILL>(a, b, want, bad, blocks);
  run_one<1, MODE, FILL>(a, b, want, bad, blocks);
  run_one<2, MODE, FILL>(a, b, want, bad, blocks);
  run_one<3, MODE, FILL>(a, b, want, bad, blocks);
  run_one<4, MODE, FILL>(a, b, want, bad, blocks);
  run_one<5, MODE, FILL>(a, b, want, bad, blocks);
  run_one<6, MODE, FILL>(a, b, want, bad, blocks);
  run_one<7, MODE, FILL>(a, b, want, bad, blocks);
  run_one<8, MODE, FILL>(a, b, want, bad, blocks);
  run_one<9, MODE, FILL>(a, b, want, bad, blocks);
  run_one<10, MODE, FILL>(a, b, want, bad, blocks);
  run_one<12, MODE, FILL>(a, b, want, bad, blocks);
}

int main() {
  // A[m][k], B[k][n] small integers (exact in bf16, exact sums): lane (j = lane&15, g = lane>>4) holds A row j,
  // k = 8g..8g+7 and B column j, k = 8g..8g+7; D lane holds rows 4g+r of column j.
  std::vector<float> A(16 * 32), B(32 * 16);
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) A[m * 32 + k] = static_cast<float>((m * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B[k * 16 + n] = static_cast<float>((n * 5 + k) % 13 - 6);
  std::vector<unsigned short> ha(64 * 8), hb(64 * 8);
  std::vector<float> hw(64 * 4);
  for (int lane = 0; lane < 64; ++lane) {
    const int j = lane & 15, g = lane >> 4;
    for (int i = 0; i < 8; ++i) { ha[lane * 8 + i] = bf16(A[j * 32 + 8 * g + i]); hb[lane * 8 + i] = bf16(B[(8 * g + i) * 16 + j]); }
    for (int r = 0; r < 4; ++r) { float s = 0.f; for (int k = 0; k < 32; ++k) s += A[(4 * g + r) * 32 + k] * B[k * 16 + j]; hw[lane * 4 + r] = s; }
  }
  s16x8 *a, *b; f32x4* want; unsigned* bad;
  HIP_OK(hipMalloc(&a, 64 * 16)); HIP_OK(hipMalloc(&b, 64 * 16)); HIP_OK(hipMalloc(&want, 64 * 16)); HIP_OK(hipMalloc(&bad, 64));
  HIP_OK(hipMemcpy(a, ha.data(), 64 * 16, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(b, hb.data(), 64 * 16, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(want, hw.data(), 64 * 16, hipMemcpyHostToDevice));
  float* sink; HIP_OK(hipMalloc(&sink, 64));
  sweep_gap<0>(a, b, bad, sink); sweep_gap<1>(a, b, bad, sink); sweep_gap<2>(a, b, bad, sink);
  sweep_gap<3>(a, b, bad, sink); sweep_gap<4>(a, b, bad, sink); sweep_gap<5>(a, b, bad, sink);
  sweep_gap<6>(a, b, bad, sink); sweep_gap<7>(a, b, bad, sink); sweep_gap<8>(a, b, bad, sink);
  if (getenv("GAP_ONLY")) return 0;
  sweep_cross<0>(a, b, bad, sink); sweep_cross<1>(a, b, bad, sink); sweep_cross<2>(a, b, bad, sink);
  sweep_cross<3>(a, b, bad, sink); sweep_cross<4>(a, b, bad, sink); sweep_cross<5>(a, b, bad, sink);
  sweep_cross<6>(a, b, bad, sink); sweep_cross<7>(a, b, bad, sink); sweep_cross<8>(a, b, bad, sink);
  if (getenv("CROSS_ONLY")) return 0;
  sweep_war<0, 0>(a, b, want, bad); sweep_war<1, 0>(a, b, want, bad); sweep_war<2, 0>(a, b, want, bad);
  sweep_war<0, 2>(a, b, want, bad); sweep_war<1, 2>(a, b, want, bad); sweep_war<2, 2>(a, b, want, bad);
  for (int blocks : {1, 2048, -1, -1024}) {   // negative: 8-wave workgroups, waves 0-3 = dense MFMA partners; one workgroup (a wave alone on its SIMD) and a full chip (several waves per SIMD)
    printf("== v_pk_add_f32 consumer\n");  sweep<0, 0>(a, b, want, bad, blocks);
    printf("== v_add_f32 consumer\n");     sweep<1, 0>(a, b, want, bad, blocks);
    printf("== v_pk_mul_f32 consumer\n");  sweep<2, 0>(a, b, want, bad, blocks);
    printf("== v_pk_fma_f32 consumer\n");  sweep<3, 0>(a, b, want, bad, blocks);
    printf("== v_pk_add_f32 consumer, one independent MFMA in between\n"); sweep<0, 1>(a, b, want, bad, blocks);
    printf("== v_add_f32 consumer, one independent MFMA in between\n");    sweep<1, 1>(a, b, want, bad, blocks);
    printf("== v_pk_add_f32 consumer, two independent MFMAs in between\n"); sweep<0, 2>(a, b, want, bad, blocks);
    printf("== v_add_f32 consumer, two independent MFMAs in between\n");    sweep<1, 2>(a, b, want, bad, blocks);
  }
  return 0;
}
