"""Stand-in for ``dgl.nn.pytorch.GATv2Conv`` restated from DGL 0.9's documented behaviour.

Only the configuration the reference constructs is supported (gnn_agents.py:93-96; drqn/agents/gnn_agents.py:17-18):
tuple in_feats, residual=True, allow_zero_in_degree=True, bias=True, share_weights=False, no dropout.
Parameter names/registration order follow DGL: attn (own Parameter) then fc_src, fc_dst, res_fc sub-modules.
Test infrastructure; NOT DGL.
"""
import math

import torch as th
import torch.nn as nn

from .. import functional as _F
from ... import function as fn


class GATv2Conv(nn.Module):
    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0., negative_slope=0.2,
                 residual=False, activation=None, allow_zero_in_degree=False, bias=True, share_weights=False):
        super().__init__()
        assert isinstance(in_feats, tuple) and residual and bias and not share_weights
        assert feat_drop == 0. and attn_drop == 0.
        self._num_heads, self._out_feats = num_heads, out_feats
        self._in_src_feats, self._in_dst_feats = in_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self.fc_src = nn.Linear(self._in_src_feats, out_feats * num_heads, bias=bias)
        self.fc_dst = nn.Linear(self._in_dst_feats, out_feats * num_heads, bias=bias)
        self.attn = nn.Parameter(th.empty(1, num_heads, out_feats))
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        self.res_fc = nn.Linear(self._in_dst_feats, num_heads * out_feats, bias=bias)
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc_src.weight, gain=gain)
        nn.init.xavier_normal_(self.fc_dst.weight, gain=gain)
        nn.init.xavier_normal_(self.attn, gain=gain)
        nn.init.constant_(self.fc_src.bias, 0)
        nn.init.constant_(self.fc_dst.bias, 0)
        nn.init.xavier_normal_(self.res_fc.weight, gain=gain)
        nn.init.constant_(self.res_fc.bias, 0)

    def forward(self, graph, feat, get_attention=False):
        with graph.local_scope():
            h_src, h_dst = feat
            feat_src = self.fc_src(h_src).view(-1, self._num_heads, self._out_feats)
            feat_dst = self.fc_dst(h_dst).view(-1, self._num_heads, self._out_feats)
            graph.srcdata.update({"el": feat_src})
            graph.dstdata.update({"er": feat_dst})
            graph.apply_edges(fn.u_add_v("el", "er", "e"))
            e = self.leaky_relu(graph.edata.pop("e"))
            e = (e * self.attn).sum(dim=-1).unsqueeze(dim=2)
            graph.edata["a"] = _F.edge_softmax(graph, e)
            graph.update_all(fn.u_mul_e("el", "a", "m"), fn.sum("m", "ft"))
            rst = graph.dstdata["ft"]
            resval = self.res_fc(h_dst).view(h_dst.shape[0], -1, self._out_feats)
            rst = rst + resval
            if self.activation:
                rst = self.activation(rst)
            return rst
