#!/usr/bin/env python
"""Package power and shader clock while ONE kernel of the C3 cycle runs back to back for a few seconds, per kernel: a bench
cycle sits at the package power limit (tools/power_probe.sh: ~1.2 kW, shader clock pulled from 2.4 to ~2.1 GHz), so what a
kernel costs the cycle is its ENERGY (power x time), not its time at full clock.  GPU box.

    python tools/power_by_kernel.py [seconds per kernel]
"""
import os
import re
import subprocess
import sys
import threading
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import _lib as L, enable_tuned_gemms, ops  # noqa: E402
from uav_bs_ctrl_amd.agents.gnn_agents import GraphObservationEncoder  # noqa: E402
import bench  # noqa: E402

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
enable_tuned_gemms()
dev = th.device("cuda")
th.manual_seed(0)
N, H = 32768, 256


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            w = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            if w and c:
                out.append((float(w.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.2)


def run(name, fn, units=1.0):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    # launches per batch so that the host stays ahead; batches until SEC seconds have passed
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    stop, samples = threading.Event(), []
    thr = threading.Thread(target=sample, args=(stop, samples))
    thr.start()
    t0 = time.perf_counter()
    n = 0
    e0.record()
    while time.perf_counter() - t0 < SEC:
        for _ in range(20):
            fn()
        n += 20
        th.cuda.synchronize()
    e1.record()
    th.cuda.synchronize()
    stop.set()
    thr.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    s = samples[len(samples) // 3:] or samples or [(0.0, 0)]      # drop the ramp
    w = sum(x for x, _ in s) / len(s)
    c = sum(y for _, y in s) / len(s)
    print(f"{name:58s} {us:9.1f} us  {w:7.0f} W  {c:6.0f} MHz  {w * us * 1e-6:8.4f} J per launch  "
          f"{w * us * 1e-6 * units:8.3f} J per cycle ({units:g} launches)", flush=True)
    return w * us * 1e-6 * units


print(f"# {SEC:.0f} s per kernel; J per cycle = J per launch x launches of that kind in one C3 cycle (50 act + update, T = 50)")
tot = 0.0
# --- K1 forward (dense), rollout size ---------------------------------------------------------------------------------
gen = th.Generator(device=dev)
gen.manual_seed(1)
g = bench.synth_batch_gpu(4096, 8, 80, "dense", dev, gen)
import types  # noqa: E402
enc = GraphObservationEncoder(dict(agent=2, ubs=2, gt=4), types.SimpleNamespace(hidden_size=H, n_heads=4, n_layers=2)).to(dev)
with th.no_grad():
    tot += run("K1 forward + f_aggr, no-grad, dense (encoder forward)", lambda: enc(g.fresh()), 50 + 51 + 50)
# --- GRU cell ---------------------------------------------------------------------------------------------------------
cell = th.nn.GRUCell(320, H).to(dev)
inp, h = th.randn(N, 320, device=dev), th.randn(N, H, device=dev)
with th.no_grad():
    tot += run("GRU cell bf16x3 forward (no-grad)", lambda: ops.gru_cell(inp, h, cell), 151)
    ops.GRU_X3_FLAGS = 1   # UAVGNN_GRU_STAGING_BLOCKS
    run("   same, staging in blocks", lambda: ops.gru_cell(inp, h, cell), 151)
    ops.GRU_X3_FLAGS = 0
    ops.GRU_X3 = False
    run("   same on fp32 MFMA (csrc/gru_fused.hip)", lambda: ops.gru_cell(inp, h, cell), 151)
    ops.GRU_X3 = True
# --- dense layers -----------------------------------------------------------------------------------------------------
x512, W = th.randn(N, 512, device=dev), th.randn(256, 512, device=dev) * 0.05
with th.no_grad():
    run("f_aggr GEMM bf16x3 [N,512] x [256,512]^T", lambda: ops.gemm_x3(x512, W), 100)
    ops.GEMM_X3_FLAGS = 4   # UAVGNN_GEMM_STAGING_INTERLEAVED
    run("   same, staging interleaved", lambda: ops.gemm_x3(x512, W), 100)
    ops.GEMM_X3_FLAGS = 0
    run("   same on the vendor fp32 GEMM", lambda: th.mm(x512, W.t()), 100)
dgi, xin = th.randn(N, 768, device=dev), th.randn(N, 320, device=dev)
tot += run("weight gradient d_gi^T inp (vendor fp32, batched split-K)", lambda: ops._wgrad(dgi, xin), 102)
Wih = th.randn(768, 320, device=dev) * 0.05
tot += run("input gradient d_gi W_ih (vendor fp32)", lambda: ops._mm_nn(dgi, Wih), 51)
print(f"# sum of the 'J per cycle' lines that are added up: {tot:.1f} J; a 120 ms cycle at 1220 W is 146 J")
