"""Independent dense formulation of the hot path's graph ops (float64).  TEST INFRASTRUCTURE ONLY.

Cross-check for ``oracle/restatement.py``: attention is evaluated on a dense [N_dst, E_src] adjacency mask with plain
``softmax`` over a masked axis instead of segment scatter/gather ops.  Shares no code with the segment
formulation, so an indexing slip in either shows up as a disagreement.
"""
import torch as th


def _mask_from_segments(seg_off, n_src):
    n = seg_off.numel() - 1
    m = th.zeros(n, n_src, dtype=th.bool)
    for v in range(n):
        m[v, int(seg_off[v]):int(seg_off[v + 1])] = True
    return m


def gatv2_dense(x_src, x_dst, seg_off, p, nh):
    """Appendix A.1 evaluated densely.  Returns [N, nh, D] after ReLU."""
    x_src, x_dst = x_src.double(), x_dst.double()
    W_s, b_s = p["fc_src.weight"].double(), p["fc_src.bias"].double()
    W_d, b_d = p["fc_dst.weight"].double(), p["fc_dst.bias"].double()
    W_r, b_r = p["res_fc.weight"].double(), p["res_fc.bias"].double()
    attn = p["attn"].double().reshape(nh, -1)
    N, E, D = x_dst.shape[0], x_src.shape[0], attn.shape[1]
    el = (x_src @ W_s.T + b_s).reshape(E, nh, D)
    er = (x_dst @ W_d.T + b_d).reshape(N, nh, D)
    z = el.unsqueeze(0) + er.unsqueeze(1)                      # [N, E, nh, D]
    lz = th.where(z > 0, z, 0.2 * z)
    e = th.einsum("nehd,hd->neh", lz, attn)                    # [N, E, nh]
    mask = _mask_from_segments(seg_off, E).unsqueeze(-1)       # [N, E, 1]
    e = e.masked_fill(~mask, -float("inf"))
    a = th.softmax(e, dim=1)
    a = th.nan_to_num(a, nan=0.0)                              # rows without in-edges
    rst = th.einsum("neh,ehd->nhd", a, el)
    rst = rst + (x_dst @ W_r.T + b_r).reshape(N, nh, D)
    return th.clamp_min(rst, 0)


def talk_attention_dense(s, q, v, talk_off, talk_src, key_size):
    """TarMAC attention (gnn_agents.py:261-267) on a dense [N_dst, N_src] multiplicity matrix."""
    s, q, v = s.double(), q.double(), v.double()
    N = s.shape[0]
    cnt = th.zeros(N, N, dtype=th.float64)                     # cnt[v, u] = multiplicity of edge u -> v
    for d in range(N):
        for e in range(int(talk_off[d]), int(talk_off[d + 1])):
            cnt[d, int(talk_src[e])] += 1
    score = (q @ s.T) / key_size                               # [dst, src]
    score = score.masked_fill(cnt == 0, -float("inf"))
    w = th.exp(score - score.max(1, keepdim=True)[0].clamp(min=-1e300)) * cnt
    w = th.nan_to_num(w, nan=0.0)
    den = w.sum(1, keepdim=True)
    a = th.where(den > 0, w / den.clamp(min=1e-300), th.zeros_like(w))
    return a @ v
