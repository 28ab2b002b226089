# Round 6: rocprofv3 kernel trace of the default bench cycle (dense) -> per-kernel and per-(kernel, grid) summaries
R=r06
cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_bench && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg --no-rho-leg --no-env-leg > /root/repo/gpurun_out/prof_bench_stdout.txt 2>&1
cd /root/repo; db=$(find gpurun_out/prof_bench -name "*results.db" | head -1)
python tools/rocprof_summary.py $db > gpurun_out/${R}_bench_kernel_stats.txt 2>&1
python tools/rocprof_by_grid.py $db > gpurun_out/${R}_bench_by_grid.txt 2>&1
rm -rf gpurun_out/prof_bench
head -45 gpurun_out/${R}_bench_by_grid.txt
