cd /root/repo
for cfg in "dense " "dense --save" "env " "env --save"; do
  set -- $cfg
  tag=$1$( [ -n "${2:-}" ] && echo save )
  bash tools/pmc.sh /root/repo/gpurun_out/pmc_$tag gatv2_hetero_fwd -- python /root/repo/tools/k1_run.py --dist $1 ${2:-} > /dev/null 2>&1
  cp gpurun_out/pmc_$tag/pmc_summary.txt gpurun_out/r04_k1_hetero_${tag}_pmc.txt
  rm -rf gpurun_out/pmc_$tag
done
# stand-alone timing: with the prepared parameter image (what learner.act / the update run) and with the in-kernel prologue
for d in env zero dense; do K1_IMAGE=1 tools/ubench/bin/k1_env_bench $d 4096 50; done > gpurun_out/r04_k1_standalone.txt 2>&1
K1_IMAGE=1 tools/ubench/bin/k1_env_bench env 208896 5 | grep "phases  *[23]:" >> gpurun_out/r04_k1_standalone.txt
echo "== in-kernel prologue (uavgnn_gatv2_hetero_fwd without an image), same box" >> gpurun_out/r04_k1_standalone.txt
for d in env dense; do tools/ubench/bin/k1_env_bench $d 4096 50 | grep "phases  *[0-3]:"; done >> gpurun_out/r04_k1_standalone.txt 2>&1
tools/ubench/bin/k1_env_bench env 208896 5 | grep "phases  *[23]:" >> gpurun_out/r04_k1_standalone.txt
echo "== rounds 1-3 kernel (pairs of destinations), same box" >> gpurun_out/r04_k1_standalone.txt
for d in env dense; do tools/ubench/bin/k1_env_bench_pair $d 4096 50 | grep "phases  *[0-3]:"; done >> gpurun_out/r04_k1_standalone.txt 2>&1
tools/ubench/bin/k1_env_bench_pair env 208896 5 | grep "phases  *[23]:" >> gpurun_out/r04_k1_standalone.txt
