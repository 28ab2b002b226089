"""Socket power and shader clock (rocm-smi, a few samples per second) while bench.py's headline cycle runs.
usage: python tools/power_trace.py [extra bench.py flags]"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-end-to-end", "--no-fp32-leg", "--no-rho-leg", "--no-env-leg",
       "--steps", "60", "--warmup", "5", *sys.argv[1:]]
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
t0 = time.time()
rows = []
while p.poll() is None:
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout
    mhz = re.findall(r"\((\d+)Mhz\)", out)
    try:
        rows.append((time.time() - t0, int(mhz[2]), float(out.strip().splitlines()[-1].split(",")[-1])))
    except (IndexError, ValueError):
        pass
r = json.loads(p.stdout.read().strip().splitlines()[-1])
for t, clk, w in rows:
    print(f"{t:7.2f} s  sclk {clk:5d} MHz  {w:7.1f} W")
busy = [x for x in rows if x[2] > 600]
if busy:
    print(f"# {len(busy)} samples above 600 W: sclk {min(b[1] for b in busy)}..{max(b[1] for b in busy)} MHz, "
          f"power {min(b[2] for b in busy):.0f}..{max(b[2] for b in busy):.0f} W, mean {sum(b[2] for b in busy) / len(busy):.0f} W")
print("# bench:", r["value"], r["unit"], f"{r['ms_per_step']:.2f} ms per cycle")
