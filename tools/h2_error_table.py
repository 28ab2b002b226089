"""gpurun_out/h2_errors.jsonl (written by the f16x2 tests of tests/test_gpu_parity.py during `pytest -m gpu`) -> a table for profiles/:
error of the f16x2 kernels against float64 beside the bf16x3 kernels' and the vendor fp32 GEMM's on the same data.

    python tools/h2_error_table.py gpurun_out/h2_errors.jsonl > profiles/r06_h2_error_tables.txt
"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/h2_errors.jsonl")]
print("# f16x2 kernels (csrc/gru_h2.hip, gemm_h2.hip, gemm_tn_h2.hip) against float64, beside the bf16x3 kernels and the vendor fp32 GEMM on the")
print("# same data.  cell: |h' - h'64| / (|h'64| + max|h'64|); GEMMs: |y - y64| / sum_k |a b| (+ |bias| + |y0|).  The condition under which")
print("# `dtype f32` stands: f16x2 not above the others (asserted by the tests that wrote these rows).")
for r in rows:
    head = {k: v for k, v in r.items() if k != "rows"}
    print(json.dumps(head))
    for e in r["rows"]:
        print(f"    {e['what']:32s} max {e['max']:.3e}   mean {e['mean']:.3e}")
