"""CPU restatement of the reference's hetero-GNN hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product (``uav_bs_ctrl_amd``) never does.

PARITY UNPINNED at the DGL boundary: the path's graph arithmetic lives in DGL 0.9.0
(``dgl=0.9.0=pypi_0``, /root/reference/requirements.txt:17), which is neither vendored under
/root/reference nor installable here, and the reference holds no test that pins a numeric result.
What this restatement *is* pinned to:
  * the reference's own wiring (algos/madrqn/agents/gnn_agents.py, dueling.py) executed UNCHANGED over
    ``oracle/dgl_standin`` in the build container -> committed fixtures ``tests/golden/*.npz``
    (generator: ``tests/golden/make_golden.py``);
  * analytical known-answer properties (tests/test_oracle_kats.py);
  * an independent dense masked-attention float64 formulation (``oracle/dense_fp64.py``).

Everything here works on plain arrays in the *segment layout* the reference's graph builder emits
(algos/madrqn/utils/env_wrappers.py:69-89): the edges of ``seen``/``near`` are grouped by destination and
``src id == edge id``, so a relation is ``x_src [E,F]`` + ``seg_off [N+1]``.  ``talk`` is CSC:
``talk_off [N+1]`` + ``talk_src [E]`` (in-edges of agent v are ``talk_src[talk_off[v]:talk_off[v+1]]``).
A general (src,dst) edge list entry point is provided for API completeness.

All functions are dtype-generic (float32 for the CPU baseline, float64 for gradcheck) and
autograd-transparent, so ``torch.autograd`` supplies the backward the HIP kernels are checked against.
"""
from __future__ import annotations

import torch as th
import torch.nn.functional as F

NEG_SLOPE = 0.2  # DGL GATv2Conv default negative_slope (gnn_agents.py:93-96 do not override it)


# --------------------------------------------------------------------------------------------------------------------
# segment helpers

def seg_ids(seg_off: th.Tensor) -> th.Tensor:
    """Destination id of every edge of a relation stored as contiguous segments."""
    n = seg_off.numel() - 1
    deg = (seg_off[1:] - seg_off[:-1]).long()
    return th.repeat_interleave(th.arange(n, device=seg_off.device), deg)


def segment_softmax(e: th.Tensor, dst: th.Tensor, n: int) -> th.Tensor:
    """``dgl.nn.functional.edge_softmax`` (norm_by='dst'): softmax of e over the in-edges of each node.

    e: [E, ...]; dst: [E] long.  Subtracts the per-destination max (as DGL does) before exp.
    """
    if e.shape[0] == 0:
        return e
    idx = dst.view(-1, *([1] * (e.dim() - 1))).expand_as(e)
    m = th.full((n,) + e.shape[1:], -float("inf"), dtype=e.dtype, device=e.device)
    m = m.scatter_reduce(0, idx, e.detach(), reduce="amax", include_self=True)
    p = th.exp(e - m.index_select(0, dst))
    den = th.zeros((n,) + e.shape[1:], dtype=e.dtype, device=e.device).index_add(0, dst, p)
    return p / den.index_select(0, dst)


def segment_sum(x: th.Tensor, dst: th.Tensor, n: int) -> th.Tensor:
    """``update_all(..., fn.sum)``: zero for nodes without in-edges."""
    return th.zeros((n,) + x.shape[1:], dtype=x.dtype, device=x.device).index_add(0, dst, x)


# --------------------------------------------------------------------------------------------------------------------
# a3: GATv2Conv((F_src, F_dst), D, nh, residual=True, allow_zero_in_degree=True, activation=ReLU)
#     constructed at gnn_agents.py:93-96 and drqn/agents/gnn_agents.py:17-18

def gatv2_conv(x_src, x_dst, src, dst, p, n_heads, activation=True):
    """General edge-list form.  p: dict with attn [1,nh,D], fc_src.{weight,bias}, fc_dst.*, res_fc.*.

    Returns [N_dst, nh, D].
    """
    n = x_dst.shape[0]
    nh = n_heads
    d = p["fc_src.weight"].shape[0] // nh
    el = F.linear(x_src, p["fc_src.weight"], p["fc_src.bias"]).view(x_src.shape[0], nh, d)
    er = F.linear(x_dst, p["fc_dst.weight"], p["fc_dst.bias"]).view(n, nh, d)
    z = el.index_select(0, src) + er.index_select(0, dst)          # fn.u_add_v
    e = (F.leaky_relu(z, NEG_SLOPE) * p["attn"]).sum(-1, keepdim=True)  # [E, nh, 1]
    a = segment_softmax(e, dst, n)
    rst = segment_sum(el.index_select(0, src) * a, dst, n)            # u_mul_e, sum
    res_b = p.get("res_fc.bias")
    rst = rst + F.linear(x_dst, p["res_fc.weight"], res_b).view(n, nh, d)
    return F.relu(rst) if activation else rst


def gatv2_conv_seg(x_src, x_dst, seg_off, p, n_heads, activation=True):
    """Segment-layout form (src id == edge id, edges grouped by destination)."""
    dst = seg_ids(seg_off)
    src = th.arange(x_src.shape[0], device=x_src.device)
    return gatv2_conv(x_src, x_dst, src, dst, p, n_heads, activation)


def sub(p: dict, prefix: str) -> dict:
    """Sub-dictionary of a state_dict: keys under ``prefix.`` with the prefix stripped."""
    k = prefix + "."
    return {n[len(k):]: v for n, v in p.items() if n.startswith(k)}


# --------------------------------------------------------------------------------------------------------------------
# a2: GraphObservationEncoder.forward (gnn_agents.py:101-107)

def graph_obs_encoder(g: dict, p: dict, n_heads: int):
    n = g["x_a"].shape[0]
    x_gt = gatv2_conv_seg(g["x_gt"], g["x_a"], g["seen_off"], sub(p, "f_conv.seen"), n_heads).reshape(n, -1)
    x_ubs = gatv2_conv_seg(g["x_ubs"], g["x_a"], g["near_off"], sub(p, "f_conv.near"), n_heads).reshape(n, -1)
    return F.relu(F.linear(th.cat((x_gt, x_ubs), 1), p["f_aggr.0.weight"], p["f_aggr.0.bias"]))


# a9: DenseObservationEncoder.forward (gnn_agents.py:62-77)
def dense_obs_encoder(x_flat, p: dict, n_layers: int):
    x = x_flat
    for l in range(n_layers):
        x = F.relu(F.linear(x, p[f"enc.{2 * l}.weight"], p[f"enc.{2 * l}.bias"]))
    return x


# --------------------------------------------------------------------------------------------------------------------
# GRUCell written out (PyTorch gate order r,z,n) so that it does not lean on ATen's fused cell.

def gru_cell(i, h, p: dict):
    gi = F.linear(i, p["weight_ih"], p["bias_ih"])
    gh = F.linear(h, p["weight_hh"], p["bias_hh"])
    H = h.shape[1]
    r = th.sigmoid(gi[:, :H] + gh[:, :H])
    z = th.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = th.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def talk_edges(g: dict):
    """(src, dst) of the talk relation from its CSC arrays."""
    return g["talk_src"].long(), seg_ids(g["talk_off"])


# a4: TarMAC.forward (gnn_agents.py:248-271)
def tarmac(g: dict, x, h, p: dict, key_size: int, n_rounds: int = 1):
    src, dst = talk_edges(g)
    n = x.shape[0]
    for _ in range(n_rounds):
        inp = th.cat((x, h.detach()), 1)
        v = F.linear(inp, p["f_val.weight"], p["f_val.bias"])
        s = F.linear(inp, p["f_sign.weight"], p["f_sign.bias"])
        q = F.linear(inp, p["f_que.weight"], p["f_que.bias"])
        e = (s.index_select(0, src) * q.index_select(0, dst)).sum(-1, keepdim=True) / key_size  # :261-262
        a = segment_softmax(e, dst, n)
        c = segment_sum(v.index_select(0, src) * a, dst, n)
        h = gru_cell(th.cat((x, c), 1), h, sub(p, "f_udt"))
    return h


def segment_mean(m, dst, n):
    """UDF reduce ``nodes.mailbox['m'].mean(1)``; nodes without in-edges keep DGL's zero fill."""
    s = segment_sum(m, dst, n)
    deg = th.zeros(n, dtype=m.dtype, device=m.device).index_add(0, dst, th.ones_like(dst, dtype=m.dtype))
    return s / deg.clamp(min=1).unsqueeze(1)


def segment_max_first(m, dst, n, key=None, return_owner=False):
    """UDF reduce ``nodes.mailbox['m'].max(1)[0]`` (gnn_agents.py:177).  torch.max(dim) routes the gradient to ONE
    index - the first maximal entry in mailbox order (= edge-id order within the in-edges of the node, which the
    stable CSC sort preserves) - so ties, which are the norm for one-hot messages, must not be split."""
    E, Fm = m.shape
    if E == 0:
        return th.zeros(n, Fm, dtype=m.dtype, device=m.device)
    idxe = dst.view(-1, 1).expand(E, Fm)
    md = m.detach() if key is None else key.detach()   # `key` decides the winner, `m` supplies the value
    mx = th.zeros(n, Fm, dtype=m.dtype, device=m.device).scatter_reduce(0, idxe, md, reduce="amax", include_self=False)
    pos = th.arange(E, device=m.device).view(-1, 1).expand(E, Fm)
    cand = th.where(md == mx.index_select(0, dst), pos, th.full_like(pos, E))
    first = th.full((n, Fm), E, dtype=th.long, device=m.device).scatter_reduce(0, idxe, cand, reduce="amin")
    has = first < E
    out = th.where(has, m.gather(0, first.clamp(max=E - 1)), th.zeros((), dtype=m.dtype, device=m.device))
    return (out, first) if return_owner else out      # first: the edge that owns (value and gradient of) each entry


# a6: BaseComm.forward (gnn_agents.py:135-148)
def base_comm(g: dict, x, h, p: dict):
    src, dst = talk_edges(g)
    node_in = th.cat((x, h.detach()), 1)
    m = F.linear(node_in, p["f_msg.weight"], p["f_msg.bias"]).index_select(0, src)
    c = segment_mean(m, dst, x.shape[0])
    return gru_cell(th.cat((x, c), 1), h, sub(p, "f_udt"))


# a6: CommNet.forward (gnn_agents.py:218-229)
def commnet(g: dict, x, h, p: dict, n_rounds: int = 1):
    src, dst = talk_edges(g)
    for _ in range(n_rounds):
        c = segment_mean(h.detach().index_select(0, src), dst, x.shape[0])
        c = F.linear(c, p["c_mod.weight"], p["c_mod.bias"])
        h = gru_cell(x + c, h, sub(p, "f_mod"))
    return h


# a6: EdgeConv.forward (gnn_agents.py:291-300)
def edge_conv(g: dict, x, h, p: dict, n_rounds: int = 1):
    src, dst = talk_edges(g)
    for _ in range(n_rounds):
        hd = h.detach()
        m_in = th.cat((x.index_select(0, src), hd.index_select(0, src),
                       x.index_select(0, dst), hd.index_select(0, dst)), 1)
        c = segment_mean(F.linear(m_in, p["f_msg.weight"], p["f_msg.bias"]), dst, x.shape[0])
        h = gru_cell(th.cat((x, c), 1), h, sub(p, "f_udt"))
    return h


# a5: DiscreteComm.forward (gnn_agents.py:180-193).  ``gumbel`` is the per-edge Gumbel(0,1) noise
# [E, msg, 2] that F.gumbel_softmax draws internally (gnn_agents.py:172); it is an explicit input here so that
# results are reproducible.  tau=0.5, hard=True, straight-through estimator.
def disc_comm(g: dict, x, h, p: dict, msg_size: int, gumbel, exact_ties: bool = False):
    """exact_ties=False reproduces the reference literally: the max (and hence the edge that receives the gradient)
    is taken over the floating-point values (y_hard - y_soft) + y_soft, so ties between several "1" bits are broken by
    rounding noise.  exact_ties=True is the exact-arithmetic rule the HIP kernel implements: the first in-edge whose
    hard bit is set (else the first in-edge) owns the channel."""
    src, dst = talk_edges(g)
    n = x.shape[0]
    logits = F.linear(th.cat((x, h.detach()), 1), p["f_enc.weight"], p["f_enc.bias"]).index_select(0, src)
    y_soft = th.softmax((logits.view(-1, msg_size, 2) + gumbel) / 0.5, -1)
    idx = y_soft.max(-1, keepdim=True)[1]
    y_hard = th.zeros_like(y_soft).scatter_(-1, idx, 1.0)
    m = (y_hard - y_soft.detach() + y_soft).flatten(1)               # [E, 2*msg]
    if exact_ties:
        c = segment_max_first(m, dst, n, key=y_hard.flatten(1))
    else:
        c = segment_max_first(m, dst, n)                              # mailbox.max(1)[0]; zero-deg stays 0
    c = F.linear(c, p["f_dec.weight"], p["f_dec.bias"])
    return gru_cell(th.cat((x, c), 1), h, sub(p, "f_udt"))


# a8: Q head (gnn_agents.py:43-46, dueling.py:13-16)
def q_head(h, p: dict, dueling: bool):
    if dueling:
        vals = F.linear(h, p["v_head.weight"], p["v_head.bias"])
        advs = F.linear(h, p["adv_head.weight"], p["adv_head.bias"])
        return vals + (advs - advs.mean(-1, keepdim=True))
    return F.linear(h, p["weight"], p["bias"])


# --------------------------------------------------------------------------------------------------------------------
# a1: GnnAgent.forward (gnn_agents.py:51-56)

def gnn_agent_forward(g: dict, h, p: dict, cfg: dict, gumbel=None):
    """g: dict of segment-layout arrays (x_gt, seen_off, x_ubs, near_off, x_a, talk_off, talk_src) or, for the
    dense encoder, ``x_flat``.  p: state_dict of the reference module.  cfg: n_heads, c, key_size, msg_size,
    n_rounds, n_layers, dueling, enc ('gnn'|'mlp').  Returns (q, h')."""
    if cfg.get("enc", "gnn") == "gnn":
        x = graph_obs_encoder(g, sub(p, "enc"), cfg["n_heads"])
    else:
        x = dense_obs_encoder(g["x_flat"], sub(p, "enc"), cfg["n_layers"])
    c = cfg["c"]
    if c is None:
        h = gru_cell(x, h, sub(p, "rnn"))
    elif c == "tarmac":
        h = tarmac(g, x, h, sub(p, "f_comm"), cfg["key_size"], cfg.get("n_rounds", 1))
    elif c == "disc":
        h = disc_comm(g, x, h, sub(p, "f_comm"), cfg["msg_size"], gumbel, cfg.get("exact_ties", False))
    elif c == "base":
        h = base_comm(g, x, h, sub(p, "f_comm"))
    elif c == "commnet":
        h = commnet(g, x, h, sub(p, "f_comm"), cfg.get("n_rounds", 1))
    elif c == "econv":
        h = edge_conv(g, x, h, sub(p, "f_comm"), cfg.get("n_rounds", 1))
    else:
        raise KeyError("Unsupported communication scheme.")
    return q_head(h, sub(p, "f_out"), cfg.get("dueling", False)), h


# DRQN twin: algos/drqn/agents/gnn_agents.py:26-30 (single relation gt -> agent, every GT connected)
def drqn_gnn_agent_forward(g: dict, h, p: dict, n_heads: int):
    n = g["x_a"].shape[0]
    x = gatv2_conv_seg(g["x_gt"], g["x_a"], g["seen_off"], sub(p, "enc"), n_heads).reshape(n, -1)
    h = gru_cell(x, h, sub(p, "rnn"))
    return F.linear(h, p["f_out.weight"], p["f_out.bias"]), h


# --------------------------------------------------------------------------------------------------------------------
# row L: the BPTT pattern of MultiAgentQLearner.update (learner.py:110-154), loss only (no optimiser).

def madrqn_loss(obs, h0, h1, acts, rews, dones, p_policy, p_target, cfg, gamma, double_q=True, next_acts=None):
    """obs: list of T+1 graph dicts; acts [T, B*n, 1] long; rews/dones broadcastable to [T, B, n].

    ``next_acts`` [T, B*n, 1] long: the double-Q action choice handed in instead of the argmax over this run's own policy
    outputs (learner.py:138 takes it from ``agent_out[1:].detach()``).  The argmax is the one discontinuous step of the loss:
    a checker that compares two precisions first verifies that the two choices differ only on rows whose top-two Q values tie
    to within the comparison's tolerance, then evaluates both sides under the SAME choice.

    Returns (loss, agent_out [T+1, N_a, A], target_out [T, N_a, A])."""
    T = len(obs) - 1
    h, h_t = h0, h1
    agent_out, target_out = [], []
    for t in range(T):
        q, h = gnn_agent_forward(obs[t], h, p_policy, cfg)
        agent_out.append(q)
        with th.no_grad():
            qn, h_t = gnn_agent_forward(obs[t + 1], h_t, p_target, cfg)
            target_out.append(qn)
    q, h = gnn_agent_forward(obs[T], h, p_policy, cfg)
    agent_out.append(q)
    agent_out, target_out = th.stack(agent_out), th.stack(target_out)
    qvals = agent_out[:-1].gather(2, acts)
    if double_q:
        if next_acts is None:
            next_acts = agent_out[1:].detach().argmax(2, keepdim=True)
        next_vals = target_out.gather(2, next_acts)
    else:
        next_vals = target_out.max(2, keepdim=True)[0]
    shp = rews.shape[:2] + (-1,)
    qvals, next_vals = qvals.view(*shp), next_vals.view(*shp)
    target = rews + gamma * (1 - dones) * next_vals
    return F.mse_loss(qvals, target.expand_as(qvals)), agent_out, target_out
