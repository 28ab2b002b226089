# Round 6: same-box A/B of the K1 forward hand-out changes (t = phase-S hand-out transposed over the workgroups, k = first destination of a
# wavefront by LDS ticket in the order the wavefronts leave phase N); checksums must agree across variants
cd /root/repo
VARS="${VARS:-t0e0 t1k0 t1k1}"
for rep in 1 2 3; do
for v in $VARS; do
  echo "== $v (pass $rep) env rollout"; K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v env 4096 50 | grep -E "phases +[0-3]:|time line|phase S done|stores drained|seen rows issued|prologue done"
done
done
for v in $VARS; do
  echo "== $v poison"; K1_POISON=1 K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v env 4096 5 | grep -E "phases +3:"
  echo "== $v no image"; tools/ubench/bin/k1_env_bench_$v env 4096 50 | grep -E "phases +3:"
  echo "== $v dense"; K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v dense 4096 50 | grep -E "phases +[0-3]:"
  echo "== $v zero"; K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v zero 4096 50 | grep -E "phases +[0-3]:"
  echo "== $v time-batched env"; K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v env 208896 5 | grep -E "phases +[0-3]:"
  echo "== $v small"; K1_POISON=1 K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v env 100 5 | grep -E "phases +3:"
  echo "== $v small2"; K1_POISON=1 K1_IMAGE=1 tools/ubench/bin/k1_env_bench_$v env 3 5 | grep -E "phases +3:"
done
