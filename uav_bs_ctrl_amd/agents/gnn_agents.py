"""Recurrent GNN agents of MADRQN on MI355X.

Interface-compatible with /root/reference/algos/madrqn/agents/gnn_agents.py (``GnnAgent(obs_shape, n_actions, args)``,
``init_hidden()``, ``forward(g, h) -> (q, h')``) and with its ``state_dict``: sub-module and
parameter names, shapes and ``parameters()`` order are identical, so a checkpoint written by either side loads into the
other and the learner's polyak ``zip(policy.parameters(), target.parameters())`` (learner.py:164) keeps working.

What differs is everything below the interface: ``g`` is a ``HeteroBatch`` (flat HBM arrays, uav_bs_ctrl_amd/graph.py)
instead of a DGLGraph, and the graph arithmetic - both GATv2 relations, the targeted attention over `talk`, the GRU gate
math - runs in hand-written HIP kernels through the C-ABI (uav_bs_ctrl_amd/ops.py -> include/uavgnn.h).
"""
from __future__ import annotations

import torch as th
import torch.nn as nn

from .. import ops
from ..graph import HeteroBatch, RelationView
from .heads import DuelingLayer


# ---------------------------------------------------------------------------------------------------------------------
class GATv2Conv(nn.Module):
    """Parameter container with DGL's GATv2Conv layout (attn, fc_src, fc_dst, res_fc) for the configuration the
    reference constructs (gnn_agents.py:93-96): tuple in_feats, residual=True, allow_zero_in_degree=True, bias=True,
    negative_slope=0.2, ReLU activation.  The arithmetic lives in K1 (csrc/gatv2.hip)."""

    def __init__(self, in_feats, out_feats, num_heads):
        super().__init__()
        self._in_src_feats, self._in_dst_feats = in_feats
        self._out_feats, self._num_heads = out_feats, num_heads
        self.attn = nn.Parameter(th.empty(1, num_heads, out_feats))
        self.fc_src = nn.Linear(self._in_src_feats, out_feats * num_heads, bias=True)
        self.fc_dst = nn.Linear(self._in_dst_feats, out_feats * num_heads, bias=True)
        self.res_fc = nn.Linear(self._in_dst_feats, out_feats * num_heads, bias=True)
        self.reset_parameters()

    def reset_parameters(self):
        """DGL's initialisation: Xavier-normal with ReLU gain on the weights and attn, zero biases (Appendix A.1)."""
        gain = nn.init.calculate_gain("relu")
        for w in (self.fc_src.weight, self.fc_dst.weight, self.attn, self.res_fc.weight):
            nn.init.xavier_normal_(w, gain=gain)
        for b in (self.fc_src.bias, self.fc_dst.bias, self.res_fc.bias):
            nn.init.zeros_(b)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # Appendix B caveat: tolerate checkpoints whose res_fc carries no bias (treated as zeros).
        key = prefix + "res_fc.bias"
        if key not in state_dict and prefix + "res_fc.weight" in state_dict:
            state_dict[key] = th.zeros_like(self.res_fc.bias)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x_src, seg_off, x_dst, order=None):
        """Single relation: [N_dst, nh, D] (used by the DRQN twin)."""
        out = ops.hetero_gatv2(x_dst, self._num_heads, [(x_src, seg_off, order, self)])
        return out.view(x_dst.shape[0], self._num_heads, self._out_feats)


class GraphObservationEncoder(nn.Module):
    """Two GATv2 relations (gt -seen-> agent, ubs -near-> agent) + Linear(2H, H) + ReLU (gnn_agents.py:80-107).
    Both relations are written by K1 into one [N_a, 2H] buffer."""

    def __init__(self, obs_shape, args):
        super().__init__()
        n_heads = args.n_heads
        out_feats = args.hidden_size
        assert out_feats % n_heads == 0, "out_feats cannot be divided by n_heads in GraphObservationLayer."
        d = out_feats // n_heads
        self._n_heads = n_heads
        self.f_conv = nn.ModuleDict({
            "seen": GATv2Conv((obs_shape["gt"], obs_shape["agent"]), d, n_heads),
            "near": GATv2Conv((obs_shape["ubs"], obs_shape["agent"]), d, n_heads),
        })
        self.f_aggr = nn.Sequential(nn.Linear(len(self.f_conv) * out_feats, out_feats), nn.ReLU())

    def forward(self, g: HeteroBatch, x=None):
        x_a = g.agent_feat()
        rels = []
        for et in ("seen", "near"):
            x_src, off = g.relation_segments(et)
            rels.append((x_src, off, g.relation_order(et), self.f_conv[et]))
        x_cat = ops.hetero_gatv2(x_a, self._n_heads, rels)                  # [N_a, 2H]
        lin = self.f_aggr[0]
        return ops.linear_relu(x_cat, lin.weight, lin.bias)


class DenseObservationEncoder(nn.Module):
    """MLP on the flattened observation (gnn_agents.py:62-77; exp2's ``o='mlp'`` with a comm block)."""

    def __init__(self, obs_shape, args):
        super().__init__()
        layers = [nn.Linear(obs_shape, args.hidden_size), nn.ReLU()]
        for _ in range(args.n_layers - 1):
            layers += [nn.Linear(args.hidden_size, args.hidden_size), nn.ReLU()]
        self.enc = nn.Sequential(*layers)

    def forward(self, g: HeteroBatch, x=None):
        return self.enc(g.agent_feat())


# ---------------------------------------------------------------------------------------------------------------------
def _gru(cell: nn.GRUCell, i_parts, h):
    """GRU cell on the concatenation of ``i_parts``: two vendor GEMMs for the [N,3H] pre-activations, gate math in the
    fused HIP kernel (K4).  The cat of a [N,H] and a [N,msg] block costs less HBM traffic than summing two [N,3H]
    partial products would."""
    inp = i_parts[0] if len(i_parts) == 1 else th.cat(i_parts, 1)
    return ops.gru_cell(inp, h, cell)     # K4: one fused launch when the shape has an instantiation


def _parent(g) -> HeteroBatch:
    return g.parent if isinstance(g, RelationView) else g


class TarMAC(nn.Module):
    """Targeted multi-agent communication (gnn_agents.py:232-271): value/signature/query projections of
    [x || stopgrad(h)], attention over in-edges scaled by 1/key_size (sic, not 1/sqrt), GRU update."""

    def __init__(self, args):
        super().__init__()
        H = args.hidden_size
        self._hidden_size, self._msg_size = H, args.msg_size
        self._key_size, self._n_rounds = args.key_size, args.n_rounds
        self.f_val = nn.Linear(2 * H, self._msg_size)
        self.f_sign = nn.Linear(2 * H, self._key_size)
        self.f_que = nn.Linear(2 * H, self._key_size)
        self.f_udt = nn.GRUCell(H + self._msg_size, H)

    def _projection_params(self):
        return ((self.f_val.weight, self.f_sign.weight, self.f_que.weight),
                (self.f_val.bias, self.f_sign.bias, self.f_que.bias))

    def fused_projection(self):
        """[f_val; f_sign; f_que] as ONE [M+2K, 2H] weight (+ bias) without a per-call ``cat``: the three Linear layers'
        parameters live back to back in one buffer - either the learner's flat parameter buffer (optim.FlatParams lays
        the members of ``_projection_params()`` out consecutively) or a private stack this method creates by re-pointing
        them (``p.data = view``) - and the stacked tensors are strided views of that memory.  Every in-place update -
        optimizer steps, ``p.data.mul_()/.add_()`` polyak averaging as the reference's learner does it
        (learner.py:165-166, which bumps no version counter), ``load_state_dict`` - is seen by construction; anything
        that REPLACES a parameter's storage (``module.to()``, ``p.data = ...``, deepcopy) breaks the adjacency, which
        the pointer check below detects, and the stack is rebuilt from the current values.  The stacked tensors carry no
        autograd history: the fused step (ops.tarmac_step) returns / sinks the gradient of the stack and splits it
        back itself."""
        ws, bs = self._projection_params()
        # fast path (151 calls per training cycle): the six data pointers are what they were when the views were made
        ptrs = (ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr())
        if getattr(self, "_stacked_ptrs", None) == ptrs:
            return self._stacked

        def adjacent(ts):
            nxt = ts[0].data_ptr()
            for t in ts:
                if t.data_ptr() != nxt or not t.is_contiguous() or t.dtype != ts[0].dtype or t.device != ts[0].device:
                    return False
                nxt += t.numel() * t.element_size()
            end = ts[0].untyped_storage().data_ptr() + ts[0].untyped_storage().nbytes()
            return nxt <= end and all(t.untyped_storage().data_ptr() == ts[0].untyped_storage().data_ptr() for t in ts)

        if not (adjacent(ws) and adjacent(bs)):
            with th.no_grad():
                Wst, bst = th.cat([w.detach() for w in ws], 0), th.cat([b.detach() for b in bs], 0)
                r = 0
                for w, b in zip(ws, bs):
                    w.data = Wst[r:r + w.shape[0]]
                    b.data = bst[r:r + b.shape[0]]
                    r += w.shape[0]
        rows = sum(w.shape[0] for w in ws)
        w0, b0 = ws[0].data, bs[0].data
        self._stacked = (w0.as_strided((rows, w0.shape[1]), (w0.shape[1], 1), w0.storage_offset()),
                         b0.as_strided((rows,), (1,), b0.storage_offset()))
        self._stacked_ptrs = tuple(t.data_ptr() for t in ws + bs)
        return self._stacked

    def forward(self, g, x, h):
        g = _parent(g)
        H, M, K = self._hidden_size, self._msg_size, self._key_size
        W = th.cat((self.f_val.weight, self.f_sign.weight, self.f_que.weight), 0)     # [M+2K, 2H]
        b = th.cat((self.f_val.bias, self.f_sign.bias, self.f_que.bias), 0)
        for _ in range(self._n_rounds):
            proj = ops.linear(x, W[:, :H], b) + ops.linear(h.detach(), W[:, H:])           # one fused projection
            c = ops.talk_attention(proj[:, M:M + K], proj[:, M + K:], proj[:, :M], g, 1.0 / K)
            h = _gru(self.f_udt, (x, c), h)
        return h


class BaseComm(nn.Module):
    """m_u = f_msg([x_u || stopgrad(h_u)]); c_v = mean over in-edges; GRU update (gnn_agents.py:113-148)."""

    def __init__(self, args):
        super().__init__()
        H = args.hidden_size
        self._hidden_size, self._msg_size = H, args.msg_size
        self.f_msg = nn.Linear(2 * H, self._msg_size)
        self.f_udt = nn.GRUCell(H + self._msg_size, H)

    def forward(self, g, x, h):
        g = _parent(g)
        H = self._hidden_size
        m = ops.linear(x, self.f_msg.weight[:, :H], self.f_msg.bias) + ops.linear(h.detach(), self.f_msg.weight[:, H:])
        c = ops.talk_attention(None, None, m, g)
        return _gru(self.f_udt, (x, c), h)


class CommNet(nn.Module):
    """c_v = mean of stopgrad(h_u); h = GRU(x + c_mod(c), h), n_rounds times (gnn_agents.py:196-229)."""

    def __init__(self, args):
        super().__init__()
        H = args.hidden_size
        self._hidden_size, self._n_rounds = H, args.n_rounds
        self.c_mod = nn.Linear(H, H)
        self.f_mod = nn.GRUCell(H, H)

    def forward(self, g, x, h):
        g = _parent(g)
        for _ in range(self._n_rounds):
            c = ops.talk_attention(None, None, h.detach().contiguous(), g)
            c = ops.linear(c, self.c_mod.weight, self.c_mod.bias)
            h = _gru(self.f_mod, (x + c,), h)
        return h


class EdgeConv(nn.Module):
    """m_uv = f_msg([x_u || sg h_u || x_v || sg h_v]); mean over in-edges; GRU (gnn_agents.py:274-300).
    The per-edge Linear is split into a source half and a destination half, so only node-level GEMMs remain:
    mean_u m_uv = mean_u(W_src in_u) + (W_dst in_v + b) [deg_v > 0]."""

    def __init__(self, args):
        super().__init__()
        H = args.hidden_size
        self._hidden_size, self._msg_size, self._n_rounds = H, args.msg_size, args.n_rounds
        self.f_msg = nn.Linear(4 * H, self._msg_size)
        self.f_udt = nn.GRUCell(H + self._msg_size, H)

    def forward(self, g, x, h):
        g = _parent(g)
        H = self._hidden_size
        W = self.f_msg.weight
        off, _ = g.talk_csc()
        has_in = (off[1:] > off[:-1]).to(x.dtype).unsqueeze(1)
        for _ in range(self._n_rounds):
            hd = h.detach()
            a_src = ops.linear(x, W[:, :H]) + ops.linear(hd, W[:, H:2 * H])
            b_dst = ops.linear(x, W[:, 2 * H:3 * H], self.f_msg.bias) + ops.linear(hd, W[:, 3 * H:])
            c = ops.talk_attention(None, None, a_src, g) + b_dst * has_in
            h = _gru(self.f_udt, (x, c), h)
        return h


class DiscreteComm(nn.Module):
    """1-bit-per-channel messages via hard Gumbel-softmax (tau = 0.5), OR-aggregation (max), decoder, GRU
    (gnn_agents.py:151-193).  The logits depend on the SOURCE node only, so they are computed per node; the Gumbel
    noise is per edge: injected (``gumbel`` [E, msg, 2] in CSC order: fixtures carrying the reference's draws) or, by
    default, drawn INSIDE K5 by a counter-based generator (Philox keyed by ``rng_state[0]``, counter = (CSC position,
    channel, ``rng_state[1]``)) - the [E, msg, 2] tensor of F.gumbel_softmax's noise never exists.  ``rng_state`` is a device
    int64 {seed, step} buffer (not part of the state_dict): seeded from torch's generator on first use, the step advances by
    one per forward (an in-place device add: graph-capturable), so a run is reproducible from torch.manual_seed."""

    def __init__(self, args):
        super().__init__()
        H = args.hidden_size
        self._hidden_size, self._msg_size = H, args.msg_size
        self.f_enc = nn.Linear(2 * H, 2 * self._msg_size)
        self.f_dec = nn.Linear(2 * self._msg_size, 2 * self._msg_size)
        self.f_udt = nn.GRUCell(H + 2 * self._msg_size, H)
        self.gumbel = None   # optional injected noise, consumed by the next forward
        self.rng_state = None

    def forward(self, g, x, h):
        g = _parent(g)
        H = self._hidden_size
        logits = ops.linear(x, self.f_enc.weight[:, :H], self.f_enc.bias) + ops.linear(h.detach(), self.f_enc.weight[:, H:])
        noise, self.gumbel = self.gumbel, None
        rng = None
        if noise is None:
            if self.rng_state is None or self.rng_state.device != x.device:
                seed = int(th.randint(0, 2 ** 62, (1,)).item())      # from torch's (seedable) default generator
                if th.distributed.is_available() and th.distributed.is_initialized():
                    # data-parallel ranks usually share torch.manual_seed: without the rank in the key every rank would draw
                    # the same noise per (edge position, channel, step)
                    seed = (seed ^ (th.distributed.get_rank() * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1)
                self.rng_state = th.tensor([seed, 0], dtype=th.int64, device=x.device)
            rng = self.rng_state
        c = ops.disc_comm_aggregate(logits, noise, g, tau=0.5, rng=rng)
        if rng is not None:
            rng[1:].add_(1)
        c = ops.linear(c, self.f_dec.weight, self.f_dec.bias)
        return _gru(self.f_udt, (x, c), h)


# ---------------------------------------------------------------------------------------------------------------------
class GnnAgent(nn.Module):
    """Recurrent agent: observation encoder -> communication block (or plain GRU) -> Q head (gnn_agents.py:12-56)."""

    def __init__(self, obs_shape, n_actions, args):
        super().__init__()
        self._hidden_size = args.hidden_size
        self._comm_protocol = args.c

        if isinstance(obs_shape, int):
            self.enc = DenseObservationEncoder(obs_shape, args)
        elif isinstance(obs_shape, dict):
            self.enc = GraphObservationEncoder(obs_shape, args)

        c = self._comm_protocol
        if c is None:
            self.rnn = nn.GRUCell(self._hidden_size, self._hidden_size)
        elif c == "base":
            self.f_comm = BaseComm(args)
        elif c == "disc":
            self.f_comm = DiscreteComm(args)
        elif c == "commnet":
            self.f_comm = CommNet(args)
        elif c == "tarmac":
            self.f_comm = TarMAC(args)
        elif c == "econv":
            self.f_comm = EdgeConv(args)
        else:
            raise KeyError("Unsupported communication scheme.")

        if args.dueling:
            self.f_out = DuelingLayer(self._hidden_size, n_actions)
        else:
            self.f_out = nn.Linear(self._hidden_size, n_actions)

    def init_hidden(self):
        return th.zeros(1, self._hidden_size)   # on CPU, as the reference does (gnn_agents.py:48-49)

    def encode(self, g: HeteroBatch):
        """Observation encoder only: x [N_a, H].  It does not depend on the hidden state, so a BPTT caller may encode
        all T+1 time steps of a sampled batch in ONE call on the time-batched graph (uav_bs_ctrl_amd.learner) - one K1
        launch per relation over (T+1) N_a destinations instead of T+1 small ones."""
        return self.enc(g, None).view(g.num_nodes("agent"), -1)

    def step(self, g: HeteroBatch, x, h, dx_out=None):
        """Communication block (or plain GRU) + Q head on pre-encoded observations x.  dx_out: optional slice of a
        ``ops.time_split`` gradient buffer that the fused step's backward writes d x into."""
        n = x.shape[0]
        if h.shape[0] != n:
            h = h.expand(n, -1)
        h = h.contiguous()
        if self._comm_protocol == "tarmac" and self.f_comm._n_rounds == 1 and isinstance(self.f_out, nn.Linear):
            return self._tarmac_step(g, x, h, dx_out)  # headline configuration: one fused autograd node per step
        if self._comm_protocol is not None:
            h = self.f_comm(g["talk"], x, h)
        else:
            h = _gru(self.rnn, (x,), h)
        if isinstance(self.f_out, DuelingLayer):
            return self.f_out(h), h
        return ops.linear(h, self.f_out.weight, self.f_out.bias), h

    def _tarmac_step(self, g, x, h, dx_out=None):
        comm = self.f_comm
        needs = th.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs and ops.GRAD_SINK is None:
            # no sink: the stacked projection weight must be part of the autograd graph
            Wp = th.cat((comm.f_val.weight, comm.f_sign.weight, comm.f_que.weight), 0)
            bp = th.cat((comm.f_val.bias, comm.f_sign.bias, comm.f_que.bias), 0)
            return ops.tarmac_step(x, h, g, comm, self.f_out, stacked=(Wp, bp), dx_out=dx_out)
        return ops.tarmac_step(x, h, g, comm, self.f_out, dx_out=dx_out)

    def forward(self, g: HeteroBatch, h):
        return self.step(g, self.encode(g), h)


class DrqnGnnAgent(nn.Module):
    """Single-agent DRQN twin (algos/drqn/agents/gnn_agents.py:9-30): one GATv2 relation gt -> agent, GRU, Linear."""

    def __init__(self, obs_shape, n_actions, args):
        super().__init__()
        self._hidden_size, self._n_heads = args.hidden_size, args.n_heads
        d = self._hidden_size // self._n_heads
        self.enc = GATv2Conv((obs_shape["gt"], obs_shape["agent"]), d, self._n_heads)
        self.rnn = nn.GRUCell(self._hidden_size, self._hidden_size)
        self.f_out = nn.Linear(self._hidden_size, n_actions)

    def init_hidden(self):
        return th.zeros(1, self._hidden_size)

    def forward(self, g: HeteroBatch, h):
        et = "seen-by" if g.has_relation("seen-by") else "seen"
        x_src, off = g.relation_segments(et)
        x = self.enc(x_src, off, g.agent_feat(), g.relation_order(et)).flatten(start_dim=1)
        h = _gru(self.rnn, (x,), h.contiguous())
        return ops.linear(h, self.f_out.weight, self.f_out.bias), h

