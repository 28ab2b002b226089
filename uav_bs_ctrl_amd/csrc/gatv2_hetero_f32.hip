// The fp32-MFMA build of the fused K1 forward (the score GEMM of rounds 1-2 on v_mfma_f32_16x16x4_f32): the same source as
// gatv2_hetero.hip compiled with K1_BF16Z = 0, reachable through uavgnn_gatv2_hetero_fwd_phases(..., phases | 256) - the A/B
// reference of the bf16-matrix-core score GEMM and the K1 part of bench.py's strict-fp32 leg (`fp32_mfma_leg`).
#define K1_BF16Z 0
#define K1_F32_TU 1
#include "gatv2_hetero_pair.inc"
