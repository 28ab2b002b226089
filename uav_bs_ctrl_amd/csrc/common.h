// Shared device helpers for libuavgnn (gfx950 / CDNA4 only: 64-lane wavefronts are hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uavgnn.h"

namespace uavgnn {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Orders LDS traffic between the lanes of ONE wavefront (LDS operations of a wave retire in issue order; this only
// stops the compiler from moving accesses across the point).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS-only variant: orders this wave's LDS traffic without draining outstanding global loads/stores (vmcnt).
__device__ __forceinline__ void wave_sync_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -static_cast<int>(e);
}

// Memory-bound grids: enough workgroups to fill 256 CUs x 8, grid-stride the rest (guide: Guideline 11).
inline int capped_grid(long long work_items, int per_block, int cap = 2048) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return static_cast<int>(b);
}

// gatv2_mfma.hip: fp32-MFMA K1 forward; returns UAVGNN_EUNSUPPORTED when (F_src, nh, D) has no MFMA instantiation.
int gatv2_fwd_mfma(int F_src, int nh, int D, const float* x_src, int E, const float* x_dst, const int32_t* seg_off,
                   const int32_t* dst_order, int N,
                   const float* W_s, const float* b_s, const float* W_d, const float* b_d, const float* attn,
                   const float* W_r, const float* b_r, float slope, float* out, int ld_out, float* a_save,
                   hipStream_t st);

// gatv2_small.hip: K1 forward for two-feature relations with mean in-degree <= 8 (nh = 4, D in {16,32,64}); returns
// UAVGNN_EUNSUPPORTED otherwise.
int gatv2_fwd_small(int F_src, int nh, int D, const float* x_src, int E, const float* x_dst, const int32_t* seg_off,
                    const int32_t* dst_order, int N, const float* W_s, const float* b_s, const float* W_d,
                    const float* b_d, const float* attn, const float* W_r, const float* b_r, float slope, float* out,
                    int ld_out, float* a_save, hipStream_t st);

}  // namespace uavgnn
