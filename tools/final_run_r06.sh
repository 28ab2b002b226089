# Round-6 measurement pass on the GPU box (everything under gpurun_out/ with the r06 prefix; cited summaries are copied to profiles/).
R=r06
set -x
cd /root/repo
rm -f gpurun_out/grad_errors.jsonl gpurun_out/h2_errors.jsonl
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${R}_final_tests.txt
python tools/grad_error_table.py gpurun_out/grad_errors.jsonl > gpurun_out/${R}_grad_errors.txt 2>&1
python tools/h2_error_table.py gpurun_out/h2_errors.jsonl > gpurun_out/${R}_h2_error_tables.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${R}_final_smoke.txt 2>&1
python bench.py > gpurun_out/${R}_final_bench_dense.json 2> gpurun_out/${R}_final_bench_dense.err
python bench.py --dist env --no-cpu-baseline --no-end-to-end > gpurun_out/${R}_final_bench_env.json 2> gpurun_out/${R}_final_bench_env.err
python bench.py --n 4 --M 40 --B 1024 --no-cpu-baseline --no-end-to-end --no-rho-leg > gpurun_out/${R}_bench_C2.json 2> gpurun_out/${R}_bench_C2.err
python bench.py --n 16 --M 200 --B 1024 --no-cpu-baseline --no-end-to-end --no-rho-leg > gpurun_out/${R}_bench_C5.json 2> gpurun_out/${R}_bench_C5.err
python tools/h2_probe.py > gpurun_out/${R}_h2_probe.txt 2>&1
python tools/gemm_tn_h2_probe.py > gpurun_out/${R}_gemm_tn_h2_probe.txt 2>&1
python tools/h2_ablate.py > gpurun_out/${R}_h2_cell_ablate.txt 2>&1
python tools/msg_probe.py > gpurun_out/${R}_final_msg_probe.txt 2>&1
python tools/gemm_h2_shape_probe.py > gpurun_out/${R}_gemm_h2_shapes_now.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench env 4096 50 > gpurun_out/${R}_k1_standalone.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench dense 4096 50 | grep "phases  *[0-3]:" >> gpurun_out/${R}_k1_standalone.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench env 208896 5 | grep "phases  *[23]:" >> gpurun_out/${R}_k1_standalone.txt 2>&1
# kernel traces: the default (dense) cycle, the D-env cycle, and the K1 rollout launch alone (kernel-only durations)
for leg in "dense " "env --dist env"; do
  set -- $leg; tag=$1; shift
  cd /tmp && export TMPDIR=/tmp && rm -rf /root/repo/gpurun_out/prof_$tag && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py "$@" --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-fp32-leg --no-rho-leg --no-env-leg > /root/repo/gpurun_out/prof_${tag}_stdout.txt 2>&1
  cd /root/repo; db=$(find gpurun_out/prof_$tag -name "*results.db" | head -1)
  python tools/rocprof_summary.py $db > gpurun_out/${R}_final_bench_${tag}_kernel_stats.txt 2>&1
  python tools/rocprof_by_grid.py $db > gpurun_out/${R}_final_bench_${tag}_by_grid.txt 2>&1
  rm -rf gpurun_out/prof_$tag
done
cd /tmp && rm -rf /root/repo/gpurun_out/prof_k1 && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_k1 -o k1 -- python /root/repo/tools/k1_run.py --dist env --reps 50 > /dev/null 2>&1
cd /root/repo; db=$(find gpurun_out/prof_k1 -name "*results.db" | head -1)
python tools/rocprof_by_grid.py $db 0 > gpurun_out/${R}_k1_env_standalone_by_grid.txt 2>&1
rm -rf gpurun_out/prof_k1
# counter passes (their own runs, no tracing): K1 forward in its four launch classes, the f16x2 cell / GEMMs, the message kernel
for cfg in "dense " "dense --save" "env " "env --save"; do
  set -- $cfg
  tag=$1$( [ -n "${2:-}" ] && echo save )
  bash tools/pmc.sh /root/repo/gpurun_out/pmc_$tag gatv2_hetero_fwd -- python /root/repo/tools/k1_run.py --dist $1 ${2:-} > /dev/null 2>&1
  cp gpurun_out/pmc_$tag/pmc_summary.txt gpurun_out/${R}_k1_hetero_${tag}_pmc.txt; rm -rf gpurun_out/pmc_$tag
done
python tools/k1_counters_json.py gpurun_out/${R}_k1_hetero_dense_pmc.txt gpurun_out/${R}_k1_hetero_densesave_pmc.txt gpurun_out/${R}_k1_hetero_env_pmc.txt gpurun_out/${R}_k1_hetero_envsave_pmc.txt > gpurun_out/${R}_k1_hetero_counters.json
bash tools/pmc.sh /root/repo/gpurun_out/pmc_h2 "h2|tarmac_msg_fwd|gemm_tn" -- python /root/repo/tools/h2_pmc_run.py > /dev/null 2>&1
cp gpurun_out/pmc_h2/pmc_summary.txt gpurun_out/${R}_h2_kernels_pmc.txt; rm -rf gpurun_out/pmc_h2
tail -2 gpurun_out/${R}_final_tests.txt
# the driver's own command line
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${R}_bench_driver_cmd.json
