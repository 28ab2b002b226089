#!/usr/bin/env python
"""torch.profiler view of ONE learner.update on the bench workload: which aten ops (with input shapes) own the device time."""
import os
import sys

import torch as th
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

dev = th.device("cuda")
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
L = MultiAgentQLearner(env_info, exp3_args("cuda"))
batch = make_sequence(4096, 8, 80, 50, "dense", dev, seed=1, distinct=4)
L.update(batch)
th.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    L.update(batch)
    th.cuda.synchronize()
def show(prof, title):
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows if not e.key.startswith(("aten::", "_", "autograd")))
    print(f"== {title}: kernel time {tot / 1e3:.2f} ms")
    for e in rows[:90]:
        print(f"{e.self_device_time_total / 1e3:9.3f} ms {e.count:5d}x {e.self_device_time_total / e.count:9.1f} us  "
              f"{e.key[:70]:70s} {str(e.input_shapes)[:90]}")


show(prof, "update")
h = L.init_hidden(4096)
_, h = L.act(batch["obs"][0], h, 0.05)
th.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof2:
    for t in range(10):
        _, h = L.act(batch["obs"][t], h, 0.05)
    th.cuda.synchronize()
show(prof2, "10 x act")
