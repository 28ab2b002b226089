"""The shipped library must not hold the instruction pattern of the gfx950 packed-fp32 / bf16-MFMA hazard
(tools/isa_audit.py, tools/ubench/mfma_pk_hazard.hip): no kernel with 16-bit-operand MFMAs may
contain a v_pk_{fma,mul,add}_f32 with an operand select.  Runs on the CPU: the code objects are disassembled."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def rows():
    import isa_audit
    from uav_bs_ctrl_amd import build
    if not os.path.exists(isa_audit.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain is not installed")
    return isa_audit.audit(build.build_lib(verbose=False))


def test_no_kernel_mixes_16_bit_mfma_with_operand_selected_packed_fp32(rows):
    exposed = [(name, ex) for status, name, _, _, _, ex in rows if status == "EXPOSED"]
    assert not exposed, f"kernels exposed to the packed-fp32 operand-select hazard: {exposed}"


def test_no_kernel_holds_the_packed_pattern_at_all(rows):
    """Round 5: the fp32 backward kernels / K5 are compiled without packed fp32 too - another wavefront's MFMA (any stream, any
    library) can no longer corrupt them, whatever the host application runs next to this library."""
    alone = [name for status, name, _, _, _, _ in rows if status == "alone"]
    assert not alone, f"kernels holding operand-selected packed fp32 instructions: {alone}"


def test_the_audit_sees_the_kernels_it_is_about(rows):
    """Guards the audit itself: it must find the matrix-core kernels (with their MFMAs), and its pattern must match what the
    compiler emits for the hazard (a line of the ROCm 7.2 disassembly of the round-4 build) - an audit that parses nothing
    passes vacuously."""
    import isa_audit
    by_name = {name: (status, n_mf, wide, n_sel) for status, name, n_mf, wide, n_sel, _ in rows}
    mf_bwd = [v for k, v in by_name.items() if "gatv2_bwd_kernelILi4ELi4ELi64ELb1ELb1" in k]
    assert len(mf_bwd) == 1 and mf_bwd[0][1] > 0 and mf_bwd[0][2] and mf_bwd[0][3] == 0, mf_bwd
    assert any("gatv2_hetero_fwd_kernel" in k and v[1] > 0 and v[2] for k, v in by_name.items())
    assert any("gru_cell_fwd_x3" in k and v[1] > 0 and v[2] for k, v in by_name.items())
    assert any("tarmac_msg_fwd_kernel" in k and v[1] > 0 and v[2] for k, v in by_name.items())
    hit = isa_audit.PK.match("\tv_pk_fma_f32 v[10:11], v[4:5], v[8:9], v[10:11] op_sel:[0,1,0]          // 000000001F40: D3B0080A")
    assert hit and "1" in hit.group(2)
    assert isa_audit.PK.match("\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]") and not isa_audit.PK.match("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7]")
    assert isa_audit.MFMA.match("\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]")
