# Round 6, GPU pass 1: the new oracle tests of the benchmarked path + a rocprofv3 kernel trace of the D-env leg (kernel-only durations)
R=r06
set -x
cd /root/repo
rm -f gpurun_out/grad_errors.jsonl
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "learner_update_at_exp3 or drqn_twin_at_exp1 or full_size_backward or graphed_cycle or graphed_update or relu_backward or gate_gradient" 2>&1 | tail -15 > gpurun_out/${R}_new_tests.txt
cat gpurun_out/${R}_new_tests.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_env && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_env -o env -- python /root/repo/bench.py --dist env --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg --no-rho-leg --no-env-leg > /root/repo/gpurun_out/prof_env_stdout.txt 2>&1
cd /root/repo; db=$(find gpurun_out/prof_env -name "*results.db" | head -1)
python tools/rocprof_by_grid.py $db > gpurun_out/${R}_env_bench_by_grid.txt 2>&1
rm -rf gpurun_out/prof_env
# the K1 rollout launch alone under the kernel trace (50 launches back to back, prepared image): kernel-only duration, no launch boundary
cd /tmp && rm -rf /root/repo/gpurun_out/prof_k1 && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_k1 -o k1 -- python /root/repo/tools/k1_run.py --dist env --reps 50 > /dev/null 2>&1
cd /root/repo; db=$(find gpurun_out/prof_k1 -name "*results.db" | head -1)
python tools/rocprof_by_grid.py $db 0 > gpurun_out/${R}_k1_env_standalone_by_grid.txt 2>&1
rm -rf gpurun_out/prof_k1
head -20 gpurun_out/${R}_env_bench_by_grid.txt; cat gpurun_out/${R}_k1_env_standalone_by_grid.txt
