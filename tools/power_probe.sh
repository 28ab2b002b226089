# Package power / shader clock sampled every 0.5 s while a command runs (GPU box):  bash tools/power_probe.sh <seconds> <command...>
# MI355X boxes of this pool cap the package near 1.2 kW: a C3 bench cycle runs AT the cap with the shader clock pulled to
# ~2.13 GHz, so a kernel variant that merely overlaps more work per cycle does not shorten the cycle - only less energy does.
N=${1:-10}; shift
( "$@" > /tmp/power_probe_cmd.log 2>&1 ) &
BP=$!
for i in $(seq 1 $((2 * N))); do
  sleep 0.5
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/.*Power (W): /W=/' -e 's/.*sclk clock level: [0-9S]*: (/sclk=/' -e 's/)//' | tr '\n' ' '
  echo
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -3 /tmp/power_probe_cmd.log
