# Round-end measurement pass on the GPU box: tests, smoke, both bench workloads, probes, rocprofv3 kernel stats of the bench,
# PMC passes (traffic + issue counters) of the K1 forward kernel.  Everything lands under gpurun_out/ with the round's prefix.
R=${1:-r03}
set -x
cd /root/repo
python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/${R}_final_tests.txt
python tools/grad_error_table.py gpurun_out/grad_errors.jsonl > gpurun_out/${R}_grad_errors.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${R}_final_smoke.txt 2>&1
python bench.py > gpurun_out/${R}_final_bench_dense.json 2> gpurun_out/${R}_final_bench_dense.err
python bench.py --dist env > gpurun_out/${R}_final_bench_env.json 2> gpurun_out/${R}_final_bench_env.err
python tools/kbench_hetero.py > gpurun_out/${R}_final_kbench_hetero.txt 2>&1
python tools/kbench.py > gpurun_out/${R}_final_kbench.txt 2>&1
for d in env zero dense; do tools/ubench/bin/k1_env_bench $d 4096 50; done > gpurun_out/${R}_final_k1_standalone.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_bench && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg --no-rho-leg --no-env-leg > /root/repo/gpurun_out/prof_bench_stdout.txt 2>&1
cd /root/repo; db=$(find gpurun_out/prof_bench -name "*results.db" | head -1)
python tools/rocprof_summary.py $db > gpurun_out/${R}_final_bench_kernel_stats.txt 2>&1
python tools/rocprof_by_grid.py $db > gpurun_out/${R}_final_bench_by_grid.txt 2>&1
rm -rf gpurun_out/prof_bench
python tools/launch_bound_probe.py > gpurun_out/${R}_final_launch_bound.txt 2>&1
python tools/launch_bound_probe.py --timers 1 2>&1 | tail -4 >> gpurun_out/${R}_final_launch_bound.txt
python tools/gru_probe.py > gpurun_out/${R}_final_gru_probe.txt 2>&1
python tools/gemm_x3_probe.py > gpurun_out/${R}_final_gemm_x3_probe.txt 2>&1
for cfg in "dense " "dense --save" "env " "env --save"; do
  set -- $cfg
  tag=$1$( [ -n "${2:-}" ] && echo save )
  bash tools/pmc.sh /root/repo/gpurun_out/pmc_$tag gatv2_hetero_fwd -- python /root/repo/tools/k1_run.py --dist $1 ${2:-} > /dev/null 2>&1
  cp gpurun_out/pmc_$tag/pmc_summary.txt gpurun_out/${R}_k1_hetero_${tag}_pmc.txt
  rm -rf gpurun_out/pmc_$tag
done
tail -2 gpurun_out/${R}_final_tests.txt
