"""QMIX monotonic mixing network (reference: algos/madrqn/agents/mixers.py:6-49).

Not graph work: two state-conditioned hyper-networks produce non-negative mixing weights, the per-agent Q-values are
mixed by two tiny batched products.  Same parameter names/shapes as the reference so checkpoints interchange.  Written
as broadcast multiply-adds over [T*B, n, embed] instead of ``bmm`` on 1 x n matrices (n <= 16: a GEMM call per sample
would be pure launch overhead)."""
import torch as th
import torch.nn as nn
import torch.nn.functional as F


class QMixer(nn.Module):
    def __init__(self, state_shape, n_agents, args):
        super().__init__()
        self.n_agents = n_agents
        self.state_dim = int(state_shape)
        self.embed_dim = args.embed_dim
        self.hyper_w_1 = nn.Linear(self.state_dim, self.embed_dim * self.n_agents)
        self.hyper_w_final = nn.Linear(self.state_dim, self.embed_dim)
        self.hyper_b_1 = nn.Linear(self.state_dim, self.embed_dim)
        self.V = nn.Sequential(nn.Linear(self.state_dim, self.embed_dim), nn.ReLU(), nn.Linear(self.embed_dim, 1))

    def forward(self, agent_qs, states):
        """agent_qs [T, B, n], states [T, B, state_dim] -> q_tot [T, B, 1]."""
        T, B = agent_qs.shape[0], agent_qs.shape[1]
        s = states.reshape(-1, self.state_dim)
        q = agent_qs.reshape(-1, self.n_agents, 1)
        w1 = self.hyper_w_1(s).abs().view(-1, self.n_agents, self.embed_dim)
        hidden = F.elu((q * w1).sum(1) + self.hyper_b_1(s))                       # [T*B, embed]
        w_final = self.hyper_w_final(s).abs()
        y = (hidden * w_final).sum(1, keepdim=True) + self.V(s)
        return y.view(T, B, 1)
