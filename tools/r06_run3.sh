cd /root/repo
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "learner_update_at_exp3 or fused_hetero or time_batched_linear or full_size" 2>&1 | grep -E "^E  |FAILED|passed|failed" | cut -c1-300 | head
for v in 1 0; do
UAVGNN_K1_ROWMAX=$v python bench.py --steps 6 --no-cpu-baseline --no-end-to-end --no-rho-leg --no-env-leg --no-fp32-leg 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K1_ROWMAX=$v', round(r['value']), round(r['ms_per_step'],2), r['loss'], r['kernel_ms_per_launch'])"
done
