# Round-5 measurement pass on the GPU box (everything under gpurun_out/ with the r05 prefix; cited summaries are copied to profiles/).
R=r05
set -x
cd /root/repo
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${R}_final_tests.txt
python tools/grad_error_table.py gpurun_out/grad_errors.jsonl > gpurun_out/${R}_grad_errors.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${R}_final_smoke.txt 2>&1
python bench.py > gpurun_out/${R}_final_bench_dense.json 2> gpurun_out/${R}_final_bench_dense.err
python bench.py --dist env --no-cpu-baseline --no-end-to-end > gpurun_out/${R}_final_bench_env.json 2> gpurun_out/${R}_final_bench_env.err
python bench.py --n 4 --M 40 --B 1024 --no-cpu-baseline --no-end-to-end --no-rho-leg > gpurun_out/${R}_bench_C2.json 2> gpurun_out/${R}_bench_C2.err
python bench.py --n 16 --M 200 --B 1024 --no-cpu-baseline --no-end-to-end --no-rho-leg > gpurun_out/${R}_bench_C5.json 2> gpurun_out/${R}_bench_C5.err
python tools/msg_probe.py > gpurun_out/${R}_final_msg_probe.txt 2>&1
python tools/cell_probe.py > gpurun_out/${R}_final_cell_probe.txt 2>&1
python tools/gemm_x3_probe.py > gpurun_out/${R}_final_gemm_x3_probe.txt 2>&1
python tools/gru_probe.py > gpurun_out/${R}_final_gru_probe.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench env 4096 50 > gpurun_out/${R}_k1_standalone.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench dense 4096 50 | grep "phases  *[0-3]:" >> gpurun_out/${R}_k1_standalone.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench env 208896 5 | grep "phases  *[23]:" >> gpurun_out/${R}_k1_standalone.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_bench && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg --no-rho-leg --no-env-leg > /root/repo/gpurun_out/prof_bench_stdout.txt 2>&1
cd /root/repo; db=$(find gpurun_out/prof_bench -name "*results.db" | head -1)
python tools/rocprof_summary.py $db > gpurun_out/${R}_final_bench_kernel_stats.txt 2>&1
python tools/rocprof_by_grid.py $db > gpurun_out/${R}_final_bench_by_grid.txt 2>&1
rm -rf gpurun_out/prof_bench
# counter passes of the new kernel of the round (the fused message launch) and of K1 forward (unchanged code; refreshed on this round's box)
bash tools/pmc.sh /root/repo/gpurun_out/pmc_msg tarmac_msg_fwd -- python /root/repo/tools/msg_probe.py > /dev/null 2>&1
cp gpurun_out/pmc_msg/pmc_summary.txt gpurun_out/${R}_msg_pmc.txt; rm -rf gpurun_out/pmc_msg
for cfg in "dense " "env "; do
  set -- $cfg
  bash tools/pmc.sh /root/repo/gpurun_out/pmc_$1 gatv2_hetero_fwd -- python /root/repo/tools/k1_run.py --dist $1 > /dev/null 2>&1
  cp gpurun_out/pmc_$1/pmc_summary.txt gpurun_out/${R}_k1_hetero_$1_pmc.txt; rm -rf gpurun_out/pmc_$1
done
tail -2 gpurun_out/${R}_final_tests.txt
