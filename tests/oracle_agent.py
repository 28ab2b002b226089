"""Test infrastructure: an nn.Module with the PARAMETER LAYOUT of uav_bs_ctrl_amd.GnnAgent whose forward is the CPU
oracle (oracle/restatement.py).  Stands in for the HIP agent wherever a test has to run without a GPU (the HIP agent has
no CPU path by design).  Never imported by the product."""
import torch.nn as nn

from oracle import restatement as R
from uav_bs_ctrl_amd import GnnAgent, HeteroBatch


class OracleBackedAgent(nn.Module):
    def __init__(self, obs_shape, n_actions, args):
        super().__init__()
        self.inner = GnnAgent(obs_shape, n_actions, args)
        self.cfg = dict(enc="gnn" if isinstance(obs_shape, dict) else "mlp", c=args.c, n_heads=args.n_heads,
                        key_size=getattr(args, "key_size", 16), msg_size=getattr(args, "msg_size", 64),
                        n_rounds=getattr(args, "n_rounds", 1), n_layers=getattr(args, "n_layers", 1),
                        dueling=args.dueling)

    def init_hidden(self):
        return self.inner.init_hidden()

    def state_dict(self, *a, **k):                        # same keys as the wrapped agent
        return self.inner.state_dict(*a, **k)

    def load_state_dict(self, sd, *a, **k):
        return self.inner.load_state_dict(sd, *a, **k)

    def forward(self, g: HeteroBatch, h):
        arrays = dict(x_a=g.agent_feat())
        if g.has_relation("seen"):
            arrays["x_gt"], arrays["seen_off"] = g.relation_segments("seen")
            arrays["x_ubs"], arrays["near_off"] = g.relation_segments("near")
        if g.has_relation("talk"):
            arrays["talk_off"], arrays["talk_src"] = g.talk_csc()
        p = dict(self.inner.named_parameters())
        return R.gnn_agent_forward(arrays, h, p, self.cfg)
