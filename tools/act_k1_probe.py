#!/usr/bin/env python
"""K1-seen launch time inside the rollout loop (HIP events of ops.KERNEL_TIMER): before any update, and after one."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd import ops  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

dev = th.device("cuda")
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
L = MultiAgentQLearner(env_info, exp3_args("cuda"))
batch = make_sequence(4096, 8, 80, 50, "dense", dev, seed=1, distinct=4)


def rollout(tag):
    ops.KERNEL_TIMER.reset(enabled=True)
    h = L.init_hidden(4096)
    for t in range(50):
        _, h = L.act(batch["obs"][t].fresh(), h, 0.05)
    s = ops.KERNEL_TIMER.summary()
    ops.KERNEL_TIMER.enabled = False
    print(tag, {k: round(v["avg_ms"] * 1e3, 1) for k, v in s.items()})


rollout("warm-up      ")
rollout("before update")
L.update(batch)
th.cuda.synchronize()
rollout("after update ")
rollout("again        ")


def loop(tag, fn, n=50):
    ops.KERNEL_TIMER.reset(enabled=True)
    with th.no_grad():
        for t in range(n):
            fn(t)
    s = ops.KERNEL_TIMER.summary()
    ops.KERNEL_TIMER.enabled = False
    print(tag, {k: round(v["avg_ms"] * 1e3, 1) for k, v in s.items()})


net = L.policy_net
obs = batch["obs"]
loop("encode only (same graph, cached order)  ", lambda t: net.encode(obs[0]))
loop("encode only (rotating graphs, cached)   ", lambda t: net.encode(obs[t % 4]))
loop("encode only (fresh graph objects)       ", lambda t: net.encode(obs[t % 4].fresh()))
h0 = L.init_hidden(4096)
xx = net.encode(obs[0]).detach()
loop("step only                               ", lambda t: net.step(obs[t % 4], xx, h0))
loop("encode + step                           ", lambda t: net.step(obs[t % 4], net.encode(obs[t % 4]), h0))
