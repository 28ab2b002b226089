cd /root/repo
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm_f16x2" 2>&1 | grep -E "^E  |FAILED|passed|failed" | cut -c1-400 | head
python bench.py --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg > gpurun_out/r06_bench_h2g.json 2> gpurun_out/r06_bench_h2g.err
UAVGNN_GEMM_H2=0 python bench.py --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg > gpurun_out/r06_bench_h2goff.json 2> gpurun_out/r06_bench_h2goff.err
UAVGNN_GEMM_X3_VARIANT=9 python bench.py --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg > gpurun_out/r06_bench_h2g_il.json 2> gpurun_out/r06_bench_h2g_il.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_h2g.json", "gpurun_out/r06_bench_h2goff.json", "gpurun_out/r06_bench_h2g_il.json"):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(r["value"]), round(r["ms_per_step"], 2), r["loss"], {k: v for k, v in r["kernel_ms_per_launch"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r06_bench_h2g.err
