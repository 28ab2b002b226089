cd /root/repo
python bench.py --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg > gpurun_out/r06_bench_h2.json 2> gpurun_out/r06_bench_h2.err
UAVGNN_GRU_H2=0 python bench.py --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg > gpurun_out/r06_bench_h2off.json 2> gpurun_out/r06_bench_h2off.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_h2.json", "gpurun_out/r06_bench_h2off.json"):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["value"], r["ms_per_step"], r["loss"], {k: v for k, v in r["kernel_ms_per_launch"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r06_bench_h2.err
UAVGNN_K1_BWD_MFMA=0 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "learner_update_at_exp3" 2>&1 | tail -8
