#!/bin/bash
# rocprofv3 PMC passes for one command (counters in their own runs, no tracing - gpurun rule).  Usage:
#   tools/pmc.sh <outdir> <kernel-regex> -- <command...>
set -u
out=$1; shift; kre=$1; shift; shift
cd /tmp; export TMPDIR=/tmp
mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -o p$i -- "$@" > "$out/p$i.log" 2>&1
done
python - "$out" "$kre" <<'PY'
import csv, glob, re, sys, collections
out, kre = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kre.search(k):
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fo:
    for k, d in acc.items():
        print("kernel:", k, file=fo)
        for c, v in sorted(d.items()):
            print(f"  {c:32s} n={len(v):3d} avg={sum(v)/len(v):.6g}", file=fo)
print(open(out + "/pmc_summary.txt").read())
PY
