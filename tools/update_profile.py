#!/usr/bin/env python
"""Target of rocprofv3 --kernel-trace: a few eager and graph-replayed updates at the reference's own sizes."""
import os, sys, time
import torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args  # noqa: E402
from uav_bs_ctrl_amd.graphs import GraphedUpdate  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402
dev = th.device("cuda")
n, M = 8, 50
L = MultiAgentQLearner(dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=50), exp3_args("cuda"))
gen = th.Generator(device=dev).manual_seed(3)
gt = th.rand(32, 51, n, M, 5, device=dev, generator=gen) * 2 - 1
gt[..., 0] = (th.rand(32, 51, n, M, device=dev, generator=gen) < 0.06).float()
ub = th.rand(32, 51, n, n - 1, 3, device=dev, generator=gen) * 2 - 1
ub[..., 0] = 1.0
m = dict(gt=gt, ubs=ub, agent=th.rand(32, 51, n, 2, device=dev), d_u2u=th.zeros(32, 51, n, n, device=dev),
         h=th.zeros(32, 51, n, 256, device=dev), act=th.randint(9, (32, 50, n), device=dev), rew=th.rand(32, 50, n, device=dev),
         done=th.zeros(32, 50, 1, device=dev))
gu = GraphedUpdate(L, 32, 50, n, M)
for _ in range(3):
    gu(m)
th.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    gu(m)
th.cuda.synchronize()
print("graphed update ms", (time.perf_counter() - t0) / 5 * 1e3)
