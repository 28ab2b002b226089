#!/usr/bin/env python
"""profiles/rNN_k1_hetero_counters.json from the PMC summaries of a round's measurement script (rocprofv3 --pmc passes of
tools/k1_run.py through tools/pmc.sh): per degree distribution and launch kind the HBM counters (KiB) and the instruction
counts of ONE launch of gatv2_hetero_fwd_kernel at C3 size.  bench.py reads it for `roofline.traffic` and `roofline.pipe_bound`.

    python tools/k1_counters_json.py profiles/r06_k1_hetero_{dense,densesave,env,envsave}_pmc.txt > profiles/r06_k1_hetero_counters.json
"""
import json
import re
import sys

_round = (re.search(r"(r\d\d)_k1_hetero", sys.argv[1]) or [None, "rNN"])[1] if len(sys.argv) > 1 else "rNN"
out = {"_comment": "ONE launch of gatv2_hetero_fwd_kernel (K1 forward, seen + near relations in one launch; the kernel of the round in the "
                   "file name: phase N on blocks of 16 destinations) on the C3 workload (B=4096, 8x80), from rocprofv3 --pmc passes collected by tools/pmc.sh "
                   "over tools/k1_run.py (counters in their own runs, FETCH_SIZE and WRITE_SIZE in separate runs, no tracing; "
                   "summaries: profiles/" + _round + "_k1_hetero_*_pmc.txt).  FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction per "
                   "MI355X_MICROARCH.md section HBM: FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled by the "
                   "reader; WRITE_SIZE taken as is.  SQ_INSTS_*: wave-instructions of the launch."}
for path in sys.argv[1:]:
    tag = re.search(r"hetero_(\w+?)_pmc", path).group(1)
    dist, kind = (tag[:-4], "training") if tag.endswith("save") else (tag, "inference")
    vals = {}
    for ln in open(path):
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+avg=([\d.e+]+)", ln)
        if m:
            vals[m.group(1)] = float(m.group(2))
    out.setdefault(dist, {})[kind] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                                      **{k: vals[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS",
                                                              "SQ_INSTS_VMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                                              "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if k in vals}}
print(json.dumps(out, indent=1))
