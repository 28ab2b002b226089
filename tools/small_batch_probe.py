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
from uav_bs_ctrl_amd import enable_tuned_gemms  # noqa: E402

print("recorded vendor-GEMM solutions:", enable_tuned_gemms() if "--no-tuned" not in sys.argv else "off")

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
    # ---- the same two calls as hipGraph replays (uav_bs_ctrl_amd/graphs.py) ----------------------------------------
    from uav_bs_ctrl_amd.graphs import GraphedAct, GraphedUpdate
    gen = th.Generator(device=dev).manual_seed(3)

    def padded(lead):
        gt = th.rand(*lead, n, M, 5, device=dev, generator=gen) * 2 - 1
        gt[..., 0] = (th.rand(*lead, n, M, device=dev, generator=gen) < 0.06).float()
        ub = th.rand(*lead, n, n - 1, 3, device=dev, generator=gen) * 2 - 1
        ub[..., 0] = 1.0
        return gt, ub, th.rand(*lead, n, 2, device=dev, generator=gen), th.zeros(*lead, n, n, device=dev)
    ga = GraphedAct(L, 1, n, M)
    o1 = [padded((1,)) for _ in range(4)]
    h = L.init_hidden(1)
    for i in range(20):
        a, h = ga(*o1[i % 4], h, 0.05)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        a, h = ga(*o1[i % 4], h, 0.05)
        a.tolist()
    dt_act_g = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for i in range(200):                          # producer writes the graph's buffers in place: replay only
        ga.h_in.copy_(h)
        a, h = ga(None, None, None, None, None, 0.05)
        a.tolist()
    dt_act_g0 = (time.perf_counter() - t0) / 200
    gu = GraphedUpdate(L, 32, 50, n, M)
    gt, ub, ag, d = padded((32, 51))
    m = dict(gt=gt, ubs=ub, agent=ag, d_u2u=d, h=th.zeros(32, 51, n, 256, device=dev),
             act=th.randint(9, (32, 50, n), device=dev), rew=th.rand(32, 50, n, device=dev), done=th.zeros(32, 50, 1, device=dev))
    for _ in range(2):
        gu(m)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gu(m)
    th.cuda.synchronize()
    dt_upd_g = (time.perf_counter() - t0) / 5
    print(f"{n} UBS x {M} GT hipGraph replays: act {dt_act_g * 1e3:.3f} ms ({dt_act_g0 * 1e3:.3f} ms with observations written in place)  |  update {dt_upd_g * 1e3:.1f} ms"
          f"  -> {32 * 50 / dt_upd_g:.0f} transitions/s   (incl. copying the inputs into the graph's buffers)")
    print(f"{n} UBS x {M} GT: act (1 env, incl. host sync) {dt_act * 1e3:.3f} ms  |  update (32 seq x T=50) {dt_upd * 1e3:.1f} ms"
          f"  -> {32 * 50 / dt_upd:.0f} transitions/s")
