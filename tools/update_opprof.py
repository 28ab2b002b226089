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
print(prof.key_averages(group_by_input_shape=True).table(sort_by="device_time_total", row_limit=45,
                                                         max_name_column_width=40, max_shapes_column_width=70))
