#!/usr/bin/env python
"""Is a bench cycle (bench.py: T act forwards + one update at C3 size) bound by the GPU or by the launch thread?  Per phase:
host time to ENQUEUE the phase (no synchronisation inside), device time between two events around it, and the wall time with a
synchronisation at the end.  enqueue ~ device means the device waits for the host in that phase: faster kernels do not
shorten it.  Kernel timers are off (their events cost host time).  GPU box.

    python tools/launch_bound_probe.py [--dist dense|env]
"""
import argparse
import gc
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from uav_bs_ctrl_amd import enable_tuned_gemms, ops  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="dense")
ap.add_argument("--B", type=int, default=4096)
ap.add_argument("--timers", type=int, default=0, help="1: keep the per-kernel event timers of bench.py on")
a = ap.parse_args()
dev = th.device("cuda", 0)
enable_tuned_gemms()
th.manual_seed(0)
n, M, T, B = 8, 80, 50, a.B
learner = MultiAgentQLearner(dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=T),
                             bench.exp3_args(str(dev)))
batch = bench.make_sequence(B, n, M, T, a.dist, dev, seed=1234, distinct=4)
ops.KERNEL_TIMER.reset(enabled=bool(a.timers))


def rollout():
    obs = [g.fresh() for g in batch["obs"]]
    h = learner.init_hidden(B)
    for t in range(T):
        _, h = learner.act(obs[t].fresh(), h, 0.05)


def update():
    obs = [g.fresh() for g in batch["obs"]]
    fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
    return learner.update(fb)


for _ in range(2):
    rollout()
    update()
th.cuda.synchronize()
gc.collect()
gc.disable()
print(f"D-{a.dist}, B = {B}: per phase  host enqueue ms | device ms (events) | wall incl. final sync ms   (kernel timers {'on' if a.timers else 'off'})")
for rep in range(3):
    line = []
    for name, fn in (("rollout (50 act)", rollout), ("update", update)):
        th.cuda.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        t1 = time.perf_counter()
        th.cuda.synchronize()
        t2 = time.perf_counter()
        line.append(f"{name}: {1e3 * (t1 - t0):6.1f} | {e0.elapsed_time(e1):6.1f} | {1e3 * (t2 - t0):6.1f}")
    print("   ".join(line))
