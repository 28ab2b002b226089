#!/usr/bin/env python
"""Is the rollout (act) loop GPU-bound?  Wall time of 50 act forwards vs the device time between two events."""
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

dev = th.device("cuda")
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
L = MultiAgentQLearner(env_info, exp3_args("cuda"))
batch = make_sequence(4096, 8, 80, 50, "dense", dev, seed=1, distinct=4)


def rollout():
    h = L.init_hidden(4096)
    for t in range(50):
        _, h = L.act(batch["obs"][t].fresh(), h, 0.05)


rollout()
th.cuda.synchronize()
for _ in range(3):
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    rollout()
    b.record()
    t_issue = time.perf_counter() - t0
    th.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    print(f"50 act: issue {1e3 * t_issue:6.1f} ms  wall {1e3 * t_wall:6.1f} ms  device span {a.elapsed_time(b):6.1f} ms")
