#!/usr/bin/env python
"""Within-process interleaved A/B of learner variants on the bench workload (guide rule 24): median / min of R rounds."""
import argparse
import os
import statistics
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dist", default="dense")
a = ap.parse_args()
dev = th.device("cuda")
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
L = MultiAgentQLearner(env_info, exp3_args("cuda"))
batch = make_sequence(4096, 8, 80, 50, a.dist, dev, seed=1, distinct=4)


def cycle():
    obs = [g.fresh() for g in batch["obs"]]      # as bench.py: derived indexes are rebuilt every cycle
    fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
    h = L.init_hidden(4096)
    for t in range(50):
        _, h = L.act(obs[t].fresh(), h, 0.05)
    L.update(fb)


def timed(fn):
    th.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    th.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


# name -> setup callable switching the variant under test (edit for the experiment at hand).
# Result of the round-1 experiment "target recurrence on a second HIP stream": 209.6 vs 211.3 ms median -> dropped.
variants = {"baseline": lambda: None, "baseline_again": lambda: None}
cycle()
res = {k: [] for k in variants}
for r in range(a.rounds):
    for k, setup in variants.items():
        setup()
        res[k].append(timed(cycle))
for k, v in res.items():
    print(f"{k:14s} median {statistics.median(v):7.1f} ms  min {min(v):7.1f}  all {[round(x, 1) for x in v]}")
