#!/usr/bin/env python
"""Audit the gfx950 machine code of the SHIPPED library for the packed-fp32 operand-select hazard measured by
tools/ubench/mfma_pk_hazard.hip (profiles/r04_mfma_pk_hazard.txt):

    a v_pk_fma_f32 / v_pk_mul_f32 whose op_sel takes the HIGH dword of src1 for the LOW result returns a wrong low result
    in lanes 48-63 when an MFMA with 128-bit operands (16x16x32 bf16 / f16, 32x32x16 bf16) is issued to the same SIMD in the
    next issue slot, by the same wave (one
    wait state in between is enough) or by another wave (no remedy but to keep the two apart).  The compiler of ROCm 7.2
    inserts no wait state for it.

A kernel is EXPOSED when it holds both a 16-bit-operand MFMA and a packed fp32 instruction with any op_sel bit set (the
audit is stricter than the measurement: src0 / src2 selects were measured clean).  A kernel without such MFMAs that holds the
packed pattern is listed as `alone`: safe only while it never shares a SIMD with a matrix-core kernel - another wavefront's MFMA
triggers the hazard too - i.e. under a one-stream calling convention.  Since round 5 the library ships NO such kernel either (the
fp32 backward kernels and K5 are compiled without packed fp32: uav_bs_ctrl_amd/build.py), and the audit fails on both classes:
the binary, not a calling convention, carries the guarantee.

    python tools/isa_audit.py [path/to/libuavgnn.so]        exit status 1 if any kernel holds the pattern (exposed or alone)

The code objects are pulled out of the fat binary with llvm-objdump --offloading and disassembled (about a second).
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LABEL = re.compile(r"^[0-9a-f]+ <(\S+)>:")
PK = re.compile(r"^\s+(v_pk_(?:fma|mul|add)_f32)\s.*\bop_sel:\[([01,]+)\]")
MFMA = re.compile(r"^\s+(v_mfma_\w+)")


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for obj in sorted(glob.glob(local + ".*hipv4-amdgcn*")):
            yield subprocess.run([OBJDUMP, "-d", obj], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def audit(lib):
    """-> list of (status, kernel, n_mfma, wide, n_select, examples)"""
    rows = []
    for text in disassemble(lib):
        name, mf, sel = None, {}, []

        def flush():
            if name and (mf or sel):
                wide = any(("bf16" in k or "f16" in k or "f8" in k or "bf8" in k) for k in mf)
                status = "EXPOSED" if (sel and wide) else ("alone" if sel else "ok")
                rows.append((status, name, sum(mf.values()), wide, len(sel), sel[:3]))

        for line in text.splitlines():
            m = LABEL.match(line)
            if m:
                flush()
                name, mf, sel = m.group(1), {}, []
                continue
            m = MFMA.match(line)
            if m:
                mf[m.group(1)] = mf.get(m.group(1), 0) + 1
            m = PK.match(line)
            if m and "1" in m.group(2):
                sel.append(line.split("//")[0].strip())
        flush()
    return rows


def demangle(name):
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        out = name
    return out.replace("uavgnn::(anonymous namespace)::", "").split("(")[0][:80]


def main(lib):
    rows = audit(lib)
    print(f"# {os.path.relpath(lib, ROOT)}: kernels holding MFMAs or packed fp32 instructions with an operand select")
    print(f"# {'status':8s} {'mfma':>5s} {'16-bit ops':>10s} {'pk op_sel':>9s}  kernel")
    for status, name, n_mf, wide, n_sel, ex in rows:
        print(f"  {status:8s} {n_mf:5d} {'yes' if wide else 'no':>10s} {n_sel:9d}  {demangle(name)}")
        for e in ex if status == "EXPOSED" else []:
            print("             ", e)
    bad = sum(r[0] == "EXPOSED" for r in rows)
    alone = sum(r[0] == "alone" for r in rows)
    print(f"# {len(rows)} kernels listed, {bad} exposed, {alone} holding the packed pattern without a 16-bit-operand MFMA of their own")
    return 1 if (bad or alone) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "uav_bs_ctrl_amd/csrc/libuavgnn.so")))
