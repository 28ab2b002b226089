// The matrix-core instantiation of the K1 backward (gatv2_bwd_kernel<4, 4, 64, true, true> of gatv2.hip), in a translation
// unit of its own because it must be compiled WITHOUT packed fp32 instructions (build.py: -target-feature -packed-fp32-ops).
//
// Measured on MI355X (tools/ubench/mfma_pk_hazard.hip, profiles/r04_mfma_pk_hazard.txt): a v_pk_fma_f32 / v_pk_mul_f32 whose
// op_sel takes the HIGH dword of src1 for the LOW result (op_sel:[0,1,0] - what the compiler emits for  lo = a.x * b.y + c.x)
// returns a wrong low result in lanes 48-63 when an MFMA with 128-bit operands (v_mfma_f32_16x16x32_bf16 / _f16,
// v_mfma_f32_32x32x16_bf16: all measured) is issued to the same SIMD in the very next issue slot - by the same wave (every time; ONE s_nop 0 between the two is enough) or by another wave of the SIMD (now and then,
// and nothing a wave can do about it).  Registers are independent: it is not a data hazard.  The fp32 MFMA 16x16x4 and the
// 64-bit-operand bf16 MFMA do not do it, other operand selects do not do it, plain v_fma_f32 does not do it.  ROCm 7.2's
// compiler knows no such hazard: the first build of this kernel staged the attention dots with exactly that instruction
// while other waves of the SIMD ran the matrix-core loop and lost the gk[1] * x[1] term of head 0 or 2 in sixteen edges of a
// destination now and then (bit-irreproducible gradients).  tools/isa_audit.py (tests/test_isa_audit.py) checks the shipped
// library for the pattern: no kernel with 16-bit-operand MFMAs may hold such an instruction.
#define UAVGNN_GATV2_BWD_MFMA_TU 1
#include "gatv2.hip"
