"""Tensor-native sequence replay - counterpart of algos/madrqn/buffer.py:7-42.

The reference keeps a deque of Python dicts holding one DGLGraph per time step and re-batches 32 x (T+1) of them with
``dgl.batch`` on every update (learner.py:99-116).  Here a fixed-length sequence of env steps is a row of a ring of
HBM-resident PADDED observation tensors (the simulator's own format, mubs_cov.py:215-242: gt [n,M,5], ubs [n,n-1,3],
agent [n,2], plus d_u2u [n,n] for the talk relation); sampling is an index gather and the graphs of a sampled batch are
rebuilt on the device by ``from_padded_obs`` (two HIP passes per time step).  Same sequence semantics as the
reference: T transitions per sequence plus the next observation / hidden state of the last one (buffer.py:26-35);
``h`` is stored per step so that ``h[0]`` / ``h[1]`` seed the policy / target BPTT (learner.py:113).

Storage per sequence at 8 x 80, T = 50:  51*8*80*5*4 B = 653 KB of GT rows (+ 4 % for the rest)  ->  5 000 sequences
(the reference's replay_size) = 3.4 GB of the 288 GB.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch as th

from .graph import HeteroBatch, from_obs_dicts, from_padded_obs, batch as hb_batch

SCHEME = ("gt", "ubs", "agent", "d_u2u", "h", "state", "act", "rew", "done")


class SequenceReplay:
    def __init__(self, capacity: int, max_seq_len: int, n_agents: int, n_gts: int, hidden_size: int,
                 n_envs: int = 1, state_dim: int = 0, r_comm: float = float("inf"), rew_dim: Optional[int] = None,
                 device="cuda"):
        T, n, M = max_seq_len, n_agents, n_gts
        self.capacity, self.T, self.n, self.M, self.n_envs = capacity, T, n, M, n_envs
        self.r_comm, self.device = r_comm, th.device(device)
        f = dict(dtype=th.float32, device=self.device)
        rd = n if rew_dim is None else rew_dim

        def ring(lead):  # committed sequences / sequences under construction (one per parallel env)
            return dict(gt=th.zeros(lead, T + 1, n, M, 5, **f), ubs=th.zeros(lead, T + 1, n, max(n - 1, 0), 3, **f),
                        agent=th.zeros(lead, T + 1, n, 2, **f), d_u2u=th.zeros(lead, T + 1, n, n, **f),
                        h=th.zeros(lead, T + 1, n, hidden_size, **f), state=th.zeros(lead, T + 1, state_dim, **f),
                        act=th.zeros(lead, T, n, dtype=th.int64, device=self.device), rew=th.zeros(lead, T, rd, **f),
                        done=th.zeros(lead, T, 1, **f))
        self.mem = ring(capacity)
        self.cur = ring(n_envs)
        self.ptr = 0            # time index inside the sequences under construction (buffer.py:16)
        self.head = 0           # next ring slot
        self.size = 0

    def __len__(self) -> int:
        return self.size

    def push(self, tr: Dict[str, th.Tensor]) -> None:
        """One transition of every parallel env.  tr: gt/ubs/agent/d_u2u/h/state [E, ...] (observation BEFORE the action),
        act [E,n], rew [E,rd], done [E,1] and the ``next_*`` observation fields (buffer.py:18-35)."""
        t = self.ptr
        for k in ("gt", "ubs", "agent", "d_u2u", "h", "state"):
            if k in tr:
                self.cur[k][:, t] = tr[k]
        for k in ("act", "rew", "done"):
            self.cur[k][:, t] = tr[k]
        self.ptr += 1
        if self.ptr == self.T:
            for k in ("gt", "ubs", "agent", "d_u2u", "h", "state"):
                if "next_" + k in tr:
                    self.cur[k][:, self.T] = tr["next_" + k]
            E = self.n_envs
            slots = (self.head + th.arange(E, device=self.device)) % self.capacity
            for k in SCHEME:
                self.mem[k][slots] = self.cur[k]
            self.head = (self.head + E) % self.capacity
            self.size = min(self.size + E, self.capacity)
            self.ptr = 0

    def stage_obs(self, tr: Dict[str, th.Tensor]) -> None:
        """The observation half of the current transition (gt/ubs/agent/d_u2u/h/state BEFORE the action), written at the
        current step WITHOUT advancing: a simulator that overwrites its observation buffers in place can be stepped before
        ``push`` receives act / rew / done / next_* - no clone of the observation in between."""
        t = self.ptr
        for k in ("gt", "ubs", "agent", "d_u2u", "h", "state"):
            if k in tr:
                self.cur[k][:, t] = tr[k]

    def sample_indices(self, batch_size: int, generator: Optional[th.Generator] = None) -> th.Tensor:
        """Without replacement, like ``random.sample`` (buffer.py:37-39)."""
        assert self.size >= batch_size, "Insufficient samples for update."
        return th.randperm(self.size, generator=generator, device=self.device)[:batch_size]

    def gather(self, idx: th.Tensor) -> Dict:
        """Batch dict in the layout ``MultiAgentQLearner.loss`` consumes: obs = list of T+1 HeteroBatch of B envs."""
        B, T, n = idx.numel(), self.T, self.n
        m = {k: v.index_select(0, idx) for k, v in self.mem.items()}
        obs = []
        for t in range(T + 1):
            if self.device.type == "cuda":
                obs.append(from_padded_obs(m["gt"][:, t], m["ubs"][:, t], m["agent"][:, t], m["d_u2u"][:, t],
                                           self.r_comm))
            else:   # host path (tests): the vectorised host builder per env
                gs = []
                for b in range(B):
                    o = [dict(agent=m["agent"][b, t, i].numpy(), ubs=m["ubs"][b, t, i].numpy(),
                              gt=m["gt"][b, t, i].numpy()) for i in range(n)]
                    gs.append(from_obs_dicts(o, m["d_u2u"][b, t].numpy(), self.r_comm))
                obs.append(hb_batch(gs))
        return dict(obs=obs, h0=m["h"][:, 0].reshape(B * n, -1), h1=m["h"][:, 1].reshape(B * n, -1),
                    acts=m["act"].permute(1, 0, 2).reshape(T, B * n, 1), rews=m["rew"].permute(1, 0, 2).contiguous(),
                    dones=m["done"].permute(1, 0, 2).contiguous(), states=m["state"].permute(1, 0, 2).contiguous())

    def sample(self, batch_size: int, generator: Optional[th.Generator] = None) -> Dict:
        return self.gather(self.sample_indices(batch_size, generator))
