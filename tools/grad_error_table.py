"""gpurun_out/grad_errors.jsonl (written by tests/util.py:grad_close during `pytest -m gpu`) -> a table for profiles/.

    python tools/grad_error_table.py gpurun_out/grad_errors.jsonl > profiles/r03_grad_errors.txt
"""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/grad_errors.jsonl")]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["test"].split("::")[-1], []).append(r)
n_rel = sum(r["decided_by"] == "1e-5" for r in rows)
n_f32 = sum(r["decided_by"].startswith("4x") for r in rows)
n_fail = sum(r["decided_by"] == "FAIL" for r in rows)
print(f"# gradient comparisons of `pytest -m gpu` against the float64 oracle: {len(rows)} tensors")
print(f"# rule: |err_i| <= max(1e-5 (max|ref| + |ref_i|), 4 x max abs error of the fp32 CPU oracle on that tensor)")
print(f"# within 1e-5 relative: {n_rel}; above 1e-5 but within 4x fp32's own absolute error: {n_f32}; failed: {n_fail}")
print(f"# rel err = max_i |err_i| / (max|ref| + |ref_i|), NO floor; cpu32 = the same for the fp32 CPU oracle\n")
print(f"{'test':100s} {'tensors':>7s} {'max rel err':>12s} {'max cpu32':>10s} {'>1e-5':>6s}")
for k, v in by.items():
    c32 = [x["cpu_fp32_rel_err"] for x in v if x["cpu_fp32_rel_err"] is not None]
    print(f"{k[:100]:100s} {len(v):7d} {max(x['rel_err'] for x in v):12.3e} {max(c32) if c32 else float('nan'):10.3e} "
          f"{sum(x['rel_err'] > 1e-5 for x in v):6d}")
over = sorted((r for r in rows if r["rel_err"] > 1e-5), key=lambda r: -r["rel_err"])
print(f"\n# every tensor above 1e-5 relative ({len(over)}):")
print(f"{'rel err':>10s} {'cpu32 rel':>10s} {'abs err':>10s} {'4x cpu32 abs':>12s} {'max|ref|':>10s}  what")
for r in over:
    c = r["cpu_fp32_rel_err"]
    print(f"{r['rel_err']:10.2e} {c if c is None else format(c, '10.2e')} {r['max_abs_err']:10.2e} {r['abs_floor']:12.2e} "
          f"{r['max_abs_ref']:10.2e}  {r['what']}  [{r['decided_by']}]")
