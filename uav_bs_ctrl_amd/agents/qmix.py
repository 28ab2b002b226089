"""QMIX monotonic mixing network (reference: algos/madrqn/agents/mixers.py:6-49).

Not graph work.  Parameter names/shapes follow the reference (``hyper_w_1``, ``hyper_w_final``, ``hyper_b_1``, ``V``)
so checkpoints interchange; the arithmetic is reorganised for the GPU: the four state-conditioned projections share
their input, so they run as ONE GEMM over the stacked weight, and the per-sample 1 x n and 1 x embed products become
broadcast multiply-reduces over [T*B, n, embed] (a ``bmm`` per sample would be pure launch overhead at n <= 16).
"""
import torch as th
import torch.nn as nn
import torch.nn.functional as F


class QMixer(nn.Module):
    def __init__(self, state_shape, n_agents, args):
        super().__init__()
        self.n_agents, self.state_dim, self.embed_dim = n_agents, int(state_shape), args.embed_dim
        self.hyper_w_1 = nn.Linear(self.state_dim, self.embed_dim * self.n_agents)
        self.hyper_w_final = nn.Linear(self.state_dim, self.embed_dim)
        self.hyper_b_1 = nn.Linear(self.state_dim, self.embed_dim)
        self.V = nn.Sequential(nn.Linear(self.state_dim, self.embed_dim), nn.ReLU(), nn.Linear(self.embed_dim, 1))

    def forward(self, agent_qs, states):
        """agent_qs [T, B, n], states [T, B, state_dim] -> q_tot [T, B, 1]."""
        T, B = agent_qs.shape[:2]
        n, e = self.n_agents, self.embed_dim
        heads = (self.hyper_w_1, self.hyper_w_final, self.hyper_b_1, self.V[0])
        proj = F.linear(states.reshape(-1, self.state_dim), th.cat([m.weight for m in heads], 0),
                        th.cat([m.bias for m in heads], 0))
        w1, w_final, b1, v_hid = proj.split((n * e, e, e, e), 1)
        hidden = F.elu((agent_qs.reshape(-1, n, 1) * w1.abs().view(-1, n, e)).sum(1) + b1)      # monotone: |w| >= 0
        v = self.V[2](F.relu(v_hid))
        return ((hidden * w_final.abs()).sum(1, keepdim=True) + v).view(T, B, 1)
