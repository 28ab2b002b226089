"""Dueling Q head (reference: algos/madrqn/agents/dueling.py:4-16).  q = V + (A - mean_a A)."""
import torch.nn as nn
from .. import ops


class DuelingLayer(nn.Module):
    def __init__(self, in_feats, n_actions):
        super().__init__()
        self.adv_head = nn.Linear(in_feats, n_actions)
        self.v_head = nn.Linear(in_feats, 1)

    def forward(self, x):
        adv = ops.linear(x, self.adv_head.weight, self.adv_head.bias)
        val = ops.linear(x, self.v_head.weight, self.v_head.bias)
        return val + adv - adv.mean(-1, keepdim=True)
