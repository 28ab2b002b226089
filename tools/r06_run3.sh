cd /root/repo
for v in 0 1 0 1; do
UAVGNN_WGRAD_OVERLAP=$v python bench.py --steps 6 --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg > gpurun_out/r06_bench_ov$v.json 2> gpurun_out/r06_bench_ov$v.err
python - <<PY
import json
r = json.loads(open("gpurun_out/r06_bench_ov$v.json").read().strip().splitlines()[-1])
print("overlap $v:", round(r["value"]), round(r["ms_per_step"], 2), r["loss"], r["params_checksum"])
PY
done
