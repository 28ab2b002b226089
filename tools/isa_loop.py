#!/usr/bin/env python
"""Inner-loop ISA listing of one kernel, as hipcc -O3 emits it for gfx950, with a static instruction mix in front - the evidence
behind "this loop is issue-bound on pipe X" statements in DESIGN.md (profiles/r04_k1_bwd_loop_isa.txt).  No GPU needed.

    python tools/isa_loop.py uav_bs_ctrl_amd/csrc/gatv2.hip gatv2_bwd_kernelILi4ELi4ELi64ELb0 [--loop N] [--min-lines 40]

The kernel is chosen by a substring of its mangled name, the loop by rank among the innermost loops (depth-maximal blocks the
assembler marks "Inner Loop Header"), longest first."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("source")
ap.add_argument("kernel")
ap.add_argument("--loop", type=int, default=0)
ap.add_argument("--defines", default="")
a = ap.parse_args()
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", f"-I{ROOT}/include",
           f"-I{ROOT}/uav_bs_ctrl_amd/csrc", *a.defines.split(), a.source, "-o", asm]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and a.kernel in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
# innermost loops: from an "Inner Loop Header" label to the last backward branch to that label
loops = []
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if not m:
        continue
    hdr = " ".join(body[i:i + 4])
    if "Inner Loop Header" not in hdr:
        continue
    lab = m.group(1)
    last = max((k for k in range(i, len(body)) if re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b", body[k]) or
                re.search(r"s_branch\s+" + re.escape(lab) + r"\b", body[k])), default=None)
    if last is not None:
        loops.append((last - i, i, last))
loops.sort(reverse=True)
if not loops:
    sys.exit("no inner loop found")
_, i0, i1 = loops[a.loop]
loop = body[i0:i1 + 1]
ops = [l.split()[0] for l in loop if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
cnt = collections.Counter(ops)


def klass(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait/nop"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_"):
        return "SALU/branch"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")):
        return "VALU (transcendental, quarter rate)"
    return "VALU"


by = collections.Counter()
for op, n in cnt.items():
    by[klass(op)] += n
print(f"# {a.source}, kernel *{a.kernel}*: innermost loop #{a.loop} by length ({len(loops)} inner loops), {len(ops)} instructions per trip")
print("# (hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only; every path of the inner branches counted once)")
print("#   " + ", ".join(f"{k} {v}" for k, v in by.most_common()))
print("#   by opcode: " + ", ".join(f"{k} {v}" for k, v in cnt.most_common()))
print()
print("\n".join(loop))
