#!/usr/bin/env python
"""Latency at the reference's own sizes: act() on ONE env graph (n agents) and update() on 32 sequences x T=50."""
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

dev = th.device("cuda")
for n, M in ((8, 50), (4, 50)):
    env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=50)
    L = MultiAgentQLearner(env_info, exp3_args("cuda"))
    one = make_sequence(1, n, M, 50, "env", dev, seed=1, distinct=4)
    b32 = make_sequence(32, n, M, 50, "env", dev, seed=2, distinct=4)
    h = L.init_hidden(1)
    for _ in range(20):
        a, h2 = L.act(one["obs"][0], h, 0.05)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        a, h = L.act(one["obs"][i % 4], h, 0.05)
        a.tolist()                                  # the reference syncs every step (learner.py:80)
    dt_act = (time.perf_counter() - t0) / 200
    for _ in range(2):
        L.update(b32)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        L.update(b32)
    th.cuda.synchronize()
    dt_upd = (time.perf_counter() - t0) / 5
    print(f"{n} UBS x {M} GT: act (1 env, incl. host sync) {dt_act * 1e3:.3f} ms  |  update (32 seq x T=50) {dt_upd * 1e3:.1f} ms"
          f"  -> {32 * 50 / dt_upd:.0f} transitions/s")
