# Round-end measurement pass on the GPU box: tests, smoke, the bench workloads (C3 dense = the default run with every leg, C3
# D-env, C2, C5 per-GPU shard), probes, rocprofv3 kernel stats of the bench, PMC passes (traffic + issue counters) of the K1
# forward and backward kernels.  Everything lands under gpurun_out/ with the round's prefix; the summaries that are cited are
# copied to profiles/ by hand.
R=${1:-r04}
set -x
cd /root/repo
python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/${R}_final_tests.txt
python tools/grad_error_table.py gpurun_out/grad_errors.jsonl > gpurun_out/${R}_grad_errors.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${R}_final_smoke.txt 2>&1
python bench.py > gpurun_out/${R}_final_bench_dense.json 2> gpurun_out/${R}_final_bench_dense.err
python bench.py --dist env --no-cpu-baseline --no-end-to-end > gpurun_out/${R}_final_bench_env.json 2> gpurun_out/${R}_final_bench_env.err
python bench.py --n 4 --M 40 --B 1024 --no-cpu-baseline --no-end-to-end --no-rho-leg > gpurun_out/${R}_bench_C2.json 2> gpurun_out/${R}_bench_C2.err
python bench.py --n 16 --M 200 --B 1024 --no-cpu-baseline --no-end-to-end --no-rho-leg > gpurun_out/${R}_bench_C5.json 2> gpurun_out/${R}_bench_C5.err
python tools/kbench_hetero.py > gpurun_out/${R}_final_kbench_hetero.txt 2>&1
python tools/kbench.py > gpurun_out/${R}_final_kbench.txt 2>&1
bash tools/r04_k1_measure.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_bench && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg --no-rho-leg --no-env-leg > /root/repo/gpurun_out/prof_bench_stdout.txt 2>&1
cd /root/repo; db=$(find gpurun_out/prof_bench -name "*results.db" | head -1)
python tools/rocprof_summary.py $db > gpurun_out/${R}_final_bench_kernel_stats.txt 2>&1
python tools/rocprof_by_grid.py $db > gpurun_out/${R}_final_bench_by_grid.txt 2>&1
rm -rf gpurun_out/prof_bench
python tools/gru_probe.py > gpurun_out/${R}_final_gru_probe.txt 2>&1
python tools/gemm_x3_probe.py > gpurun_out/${R}_final_gemm_x3_probe.txt 2>&1
python tools/gemm_tn_big_probe.py > gpurun_out/${R}_gemm_tn_big_probe.txt 2>&1
# K1 backward `seen`: counter passes of the kernel bench, matrix-core kernel (default) and packed-FMA kernel
bash tools/pmc.sh /root/repo/gpurun_out/pmc_bwd gatv2_bwd_kernel -- python /root/repo/tools/kbench.py --dists dense --reps 2 > /dev/null 2>&1
cp gpurun_out/pmc_bwd/pmc_summary.txt gpurun_out/${R}_k1_bwd_mfma_pmc.txt; rm -rf gpurun_out/pmc_bwd
UAVGNN_K1_BWD_MFMA=0 bash tools/pmc.sh /root/repo/gpurun_out/pmc_bwd gatv2_bwd_kernel -- python /root/repo/tools/kbench.py --dists dense --reps 2 > /dev/null 2>&1
cp gpurun_out/pmc_bwd/pmc_summary.txt gpurun_out/${R}_k1_bwd_pmc.txt; rm -rf gpurun_out/pmc_bwd
# the packed-fp32 operand-select / bf16-MFMA hazard in isolation, and the K1 backward's repeatability at sizes with two workgroups per CU
tools/ubench/bin/mfma_pk_hazard 2>&1 | python tools/hazard_report.py > gpurun_out/${R}_mfma_pk_hazard.txt
REPS=100 REL=1 CASES="[(4096,64,64),(8192,100,128),(8192,16,130),(32768,20,60)]" python tools/k1_bwd_repro.py > gpurun_out/${R}_k1_bwd_repro.txt 2>&1
tail -2 gpurun_out/${R}_final_tests.txt
