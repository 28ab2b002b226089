#!/usr/bin/env python
"""Bitwise reproducibility of one update at the bench shapes (with whatever GEMM solutions are selected): two learners
from the same seed, same batch -> identical loss and parameters after the step."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner, params_checksum  # noqa: E402
import torch.cuda.tunable as tun  # noqa: E402

env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
batch = make_sequence(4096, 8, 80, 50, "dense", th.device("cuda"), seed=1, distinct=4)
out = []
for rep in range(2):
    th.manual_seed(0)
    L = MultiAgentQLearner(env_info, exp3_args("cuda"))
    r = L.update(batch)
    out.append((float(r["LossQ"]), params_checksum(L.policy_net).tolist()))
print("tunable enabled:", tun.is_enabled(), "tuning:", tun.tuning_is_enabled(), "results:", len(tun.get_results()))
print(out[0])
print(out[1])
print("bitwise reproducible:", out[0] == out[1])
