set -x
cd /root/repo
python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r02_final_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_final_smoke.txt 2>&1
python bench.py > gpurun_out/r02_final_bench_dense.json 2> gpurun_out/r02_final_bench_dense.err
python bench.py --dist env > gpurun_out/r02_final_bench_env.json 2> gpurun_out/r02_final_bench_env.err
python tools/gru_probe.py > gpurun_out/r02_gru_probe.txt 2>&1
python tools/small_batch_probe.py > gpurun_out/r02_small_batch.txt 2>&1
python tools/kbench_hetero.py > gpurun_out/r02_final_kbench_hetero.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_bench && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg > /root/repo/gpurun_out/prof_bench_stdout.txt 2>&1
cd /root/repo; python tools/rocprof_summary.py $(ls gpurun_out/prof_bench/*results.db | head -1) > gpurun_out/r02_final_bench_kernel_stats.txt 2>&1
tail -2 gpurun_out/r02_final_tests.txt
