"""Q heads of the agent.

``DuelingLayer`` keeps the reference's parameter layout (``adv_head``, ``v_head``; algos/madrqn/agents/dueling.py:4-16)
so checkpoints interchange, but evaluates both heads with ONE GEMM over the stacked [n_actions + 1, H] weight:
q = V + (A - mean_a A).
"""
import torch as th
import torch.nn as nn

from .. import ops


class DuelingLayer(nn.Module):
    def __init__(self, in_feats, n_actions):
        super().__init__()
        self.n_actions = n_actions
        self.adv_head = nn.Linear(in_feats, n_actions)
        self.v_head = nn.Linear(in_feats, 1)

    def forward(self, x):
        w = th.cat((self.adv_head.weight, self.v_head.weight), 0)
        b = th.cat((self.adv_head.bias, self.v_head.bias), 0)
        out = ops.linear(x, w, b)                                  # [N, A + 1]: advantages | state value
        adv, val = out[:, :self.n_actions], out[:, self.n_actions:]
        return val + adv - adv.mean(-1, keepdim=True)
