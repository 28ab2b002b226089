#!/usr/bin/env python
"""Where the simulator-inclusive cycle (bench.end_to_end) spends its time: per-phase wall clock with a device sync after
each phase (phases are >= 0.1 ms, the sync cost is noise).  GPU box."""
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args  # noqa: E402
from uav_bs_ctrl_amd import enable_tuned_gemms, from_padded_obs  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402
from uav_bs_ctrl_amd.replay import SequenceReplay  # noqa: E402
from uav_bs_ctrl_amd.sim import BatchedUbsCoverageEnv, MapParams  # noqa: E402

enable_tuned_gemms()
dev = th.device("cuda")
B, n, M, T = 4096, 8, 80, 50
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=T)
learner = MultiAgentQLearner(env_info, exp3_args("cuda"))
mp = MapParams(n_ubs=n, n_gts=M, n_rbs=5, range_pos=6000.0, episode_limit=T, dt=40.0, r_cov=100.0, r_sns=400.0,
               vels=(5.0, 10.0), n_dirs=4, reward_scale_rate=10.0)
env = BatchedUbsCoverageEnv(mp, B, dev)
rb = SequenceReplay(capacity=B, max_seq_len=T, n_agents=n, n_gts=M, hidden_size=256, n_envs=B, state_dim=0, r_comm=mp.r_comm,
                    device=dev)
gen = th.Generator(device=dev).manual_seed(99)
acc = {}


class phase:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        th.cuda.synchronize()
        self.t = time.perf_counter()

    def __exit__(self, *a):
        th.cuda.synchronize()
        acc[self.name] = acc.get(self.name, 0.0) + time.perf_counter() - self.t


def cycle():
    grid = 200.0
    spot = th.randint(0, 26, (B, 1, 2), device=dev, generator=gen).double() * grid
    grp = spot + th.randint(0, 4, (B, M // 5, 2), device=dev, generator=gen).double() * grid
    gts = (grp.repeat_interleave(5, 1) + 100.0 * (th.rand(B, M, 2, device=dev, generator=gen, dtype=th.float64) - 0.5)).clamp(0, mp.range_pos).float()
    ubs = th.randint(0, 30, (B, n, 2), device=dev, generator=gen).double() * grid
    with phase("reset"):
        o = env.reset(ubs, gts, generator=gen)
    h = learner.init_hidden(B)
    for t in range(T):
        with phase("rollout: graph construction (static)"):
            g = from_padded_obs(o["gt"], o["ubs"], o["agent"], o["d_u2u"], r_comm=mp.r_comm, static=True)
        with phase("rollout: clone observations"):
            cur = {k: o[k].clone() for k in ("gt", "ubs", "agent", "d_u2u")}
        with phase("rollout: act"):
            acts, h2 = learner.act(g, h, 0.05)
        with phase("rollout: simulator step"):
            o, rew, done, _ = env.step(acts)
        with phase("rollout: replay push"):
            rb.push(dict(cur, h=h.view(B, n, -1), state=th.zeros(B, 0, device=dev), act=acts.view(B, n), rew=rew.float(),
                         done=th.zeros(B, 1, device=dev), next_gt=o["gt"], next_ubs=o["ubs"], next_agent=o["agent"],
                         next_d_u2u=o["d_u2u"], next_h=h2.view(B, n, -1), next_state=th.zeros(B, 0, device=dev)))
        h = h2
    m = rb.mem
    with phase("update: replay -> time-major"):
        tm = {k: m[k].transpose(0, 1).contiguous() for k in ("gt", "ubs", "agent", "d_u2u")}
    with phase("update: 51 per-step graphs (static)"):
        obs = [from_padded_obs(tm["gt"][t], tm["ubs"][t], tm["agent"][t], tm["d_u2u"][t], r_comm=mp.r_comm, static=True) for t in range(T + 1)]
    flat = lambda x, lo: x[lo:].reshape((-1,) + x.shape[2:])  # noqa: E731
    with phase("update: 2 time-batched graphs"):
        oa = from_padded_obs(flat(tm["gt"], 0), flat(tm["ubs"], 0), flat(tm["agent"], 0))
        oan = from_padded_obs(flat(tm["gt"], 1), flat(tm["ubs"], 1), flat(tm["agent"], 1))
    batch = dict(obs=obs, obs_all=oa, obs_all_next=oan, h0=m["h"][:, 0].reshape(B * n, -1), h1=m["h"][:, 1].reshape(B * n, -1),
                 acts=m["act"].permute(1, 0, 2).reshape(T, B * n, 1), rews=m["rew"].permute(1, 0, 2).contiguous(),
                 dones=m["done"].permute(1, 0, 2).contiguous())
    with phase("update: learner.update"):
        learner.update(batch)


cycle()
acc.clear()
for _ in range(2):
    cycle()
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:45s} {1e3 * v / 2:8.2f} ms per cycle  {100 * v / tot:5.1f} %")
print(f"{'sum (with a sync after every phase)':45s} {1e3 * tot / 2:8.2f} ms")
