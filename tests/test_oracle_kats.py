"""Analytical known-answer tests of the CPU restatement - independent of any DGL knowledge - plus the
dense float64 cross-check and gradcheck."""
import math

import pytest
import torch as th
import torch.nn.functional as F

from oracle import dense_fp64 as Dn
from oracle import restatement as R
from oracle.closed_form import closed_form_tensor as cf
from tests.util import assert_close

NH, D, FS = 4, 8, 4


@pytest.fixture(autouse=True)
def _float64_default():
    """float64 for this module only (a process-wide default would leak into the GPU tests collected alongside)."""
    old = th.get_default_dtype()
    th.set_default_dtype(th.float64)
    yield
    th.set_default_dtype(old)


def gat_params(fs=FS, nh=NH, d=D, seed=0.0):
    H = nh * d
    return {"attn": cf((1, nh, d), 1.0 + seed, 0.7), "fc_src.weight": cf((H, fs), 2.0 + seed, 0.5),
            "fc_src.bias": cf((H,), 3.0 + seed, 0.2), "fc_dst.weight": cf((H, 2), 4.0 + seed, 0.5),
            "fc_dst.bias": cf((H,), 5.0 + seed, 0.2), "res_fc.weight": cf((H, 2), 6.0 + seed, 0.5),
            "res_fc.bias": cf((H,), 7.0 + seed, 0.2)}


def ragged(degs, fs=FS, seed=0.0):
    off = th.tensor([0] + list(degs)).cumsum(0).int()
    E = int(off[-1])
    return cf((E, fs), 8.0 + seed, 1.0), cf((len(degs), 2), 9.0 + seed, 1.0).abs(), off


def test_kat_degree0_is_residual_only():
    p = gat_params()
    x_s, x_d, off = ragged([0, 0, 0])
    out = R.gatv2_conv_seg(x_s, x_d, off, p, NH).reshape(3, -1)
    assert_close(out, F.relu(F.linear(x_d, p["res_fc.weight"], p["res_fc.bias"])), 1e-14)


def test_kat_degree1_independent_of_attn_and_fc_dst():
    p = gat_params()
    x_s, x_d, off = ragged([1, 1, 1, 1])
    out = R.gatv2_conv_seg(x_s, x_d, off, p, NH).reshape(4, -1)
    ref = F.relu(F.linear(x_s, p["fc_src.weight"], p["fc_src.bias"]) + F.linear(x_d, p["res_fc.weight"], p["res_fc.bias"]))
    assert_close(out, ref, 1e-14)
    p2 = dict(p, attn=p["attn"] * -3 + 1, **{"fc_dst.weight": p["fc_dst.weight"] * 2, "fc_dst.bias": p["fc_dst.bias"] - 1})
    assert_close(R.gatv2_conv_seg(x_s, x_d, off, p2, NH).reshape(4, -1), ref, 1e-14)


def test_kat_zero_attn_is_uniform_mean():
    p = gat_params()
    p["attn"] = th.zeros_like(p["attn"])
    x_s, x_d, off = ragged([3, 5, 0, 2])
    out = R.gatv2_conv_seg(x_s, x_d, off, p, NH).reshape(4, -1)
    el = F.linear(x_s, p["fc_src.weight"], p["fc_src.bias"])
    dst = R.seg_ids(off)
    mean = R.segment_mean(el, dst, 4)
    assert_close(out, F.relu(mean + F.linear(x_d, p["res_fc.weight"], p["res_fc.bias"])), 1e-13)


def test_kat_identical_neighbours_uniform():
    p = gat_params()
    x_s, x_d, off = ragged([6])
    x_s = x_s[:1].expand(6, -1).contiguous()
    out = R.gatv2_conv_seg(x_s, x_d, off, p, NH).reshape(1, -1)
    ref = F.relu(F.linear(x_s[:1], p["fc_src.weight"], p["fc_src.bias"]) + F.linear(x_d, p["res_fc.weight"], p["res_fc.bias"]))
    assert_close(out, ref, 1e-13)


def test_kat_edge_order_permutation_invariance():
    p = gat_params()
    x_s, x_d, off = ragged([7, 0, 4])
    out = R.gatv2_conv_seg(x_s, x_d, off, p, NH)
    perm = th.cat([th.randperm(7, generator=th.Generator().manual_seed(1)), th.tensor([7, 8, 9, 10]).flip(0)])
    assert_close(R.gatv2_conv_seg(x_s[perm], x_d, off, p, NH), out, 1e-13)


def test_kat_batch_equivariance():
    p = gat_params()
    xs1, xd1, o1 = ragged([2, 0, 5], seed=0.3)
    xs2, xd2, o2 = ragged([1, 9], seed=0.6)
    o12 = th.cat([o1, o2[1:] + o1[-1]])
    both = R.gatv2_conv_seg(th.cat([xs1, xs2]), th.cat([xd1, xd2]), o12, p, NH)
    sep = th.cat([R.gatv2_conv_seg(xs1, xd1, o1, p, NH), R.gatv2_conv_seg(xs2, xd2, o2, p, NH)])
    assert_close(both, sep, 1e-14)


def test_kat_lrelu_abs_identity_used_by_kernels():
    """A.3(i): lrelu_0.2(z) = 0.6 z + 0.4 |z| - the algebra the HIP kernel relies on."""
    z = cf((50, 7), 0.1, 3.0)
    assert_close(0.6 * z + 0.4 * z.abs(), F.leaky_relu(z, 0.2), 1e-15)


def test_kat_input_space_aggregation():
    """A.3(ii): sum_u a_uv el[u] = W_s (sum_u a_uv x_u) + b_s [deg>0]."""
    p = gat_params()
    x_s, x_d, off = ragged([5, 0, 3])
    dst = R.seg_ids(off)
    el = F.linear(x_s, p["fc_src.weight"], p["fc_src.bias"]).view(-1, NH, D)
    er = F.linear(x_d, p["fc_dst.weight"], p["fc_dst.bias"]).view(-1, NH, D)
    e = (F.leaky_relu(el + er[dst], 0.2) * p["attn"]).sum(-1, keepdim=True)
    a = R.segment_softmax(e, dst, 3)                                # [E, nh, 1]
    lhs = R.segment_sum(el * a, dst, 3)
    s = R.segment_sum(a * x_s.unsqueeze(1), dst, 3)                 # [N, nh, F]
    W = p["fc_src.weight"].view(NH, D, FS)
    has = (off[1:] > off[:-1]).double().view(-1, 1, 1)
    rhs = th.einsum("nhf,hdf->nhd", s, W) + p["fc_src.bias"].view(1, NH, D) * has
    assert_close(lhs, rhs, 1e-13)


@pytest.mark.parametrize("degs", [[3, 0, 1, 17, 64], [80, 2]])
def test_dense_fp64_cross_check_gatv2(degs):
    p = gat_params()
    x_s, x_d, off = ragged(degs)
    assert_close(R.gatv2_conv_seg(x_s, x_d, off, p, NH), Dn.gatv2_dense(x_s, x_d, off, p, NH), 1e-12)


def _talk(n_list, mode):
    """CSC talk graph for envs of sizes n_list: 'complete' | 'self' | 'ring'."""
    off, src, base = [0], [], 0
    for n in n_list:
        for v in range(n):
            if mode == "complete":
                ins = list(range(n))
            elif mode == "self":
                ins = [v]
            else:
                ins = sorted({v, (v + 1) % n})
            src += [base + u for u in ins]
            off.append(len(src))
        base += n
    return th.tensor(off, dtype=th.int32), th.tensor(src, dtype=th.int32)


def tarmac_params(H=16, msg=8, key=4):
    return {"f_val.weight": cf((msg, 2 * H), 1.5, 0.3), "f_val.bias": cf((msg,), 2.5, 0.1),
            "f_sign.weight": cf((key, 2 * H), 3.5, 0.3), "f_sign.bias": cf((key,), 4.5, 0.1),
            "f_que.weight": cf((key, 2 * H), 5.5, 0.3), "f_que.bias": cf((key,), 6.5, 0.1),
            "f_udt.weight_ih": cf((3 * H, H + msg), 7.5, 0.3), "f_udt.weight_hh": cf((3 * H, H), 8.5, 0.3),
            "f_udt.bias_ih": cf((3 * H,), 9.5, 0.1), "f_udt.bias_hh": cf((3 * H,), 10.5, 0.1)}


def test_kat_tarmac_zero_signature_is_mean_and_selfloop_is_identity():
    H, msg, key = 16, 8, 4
    p = tarmac_params(H, msg, key)
    x, h = cf((7, H), 0.2, 1.0), cf((7, H), 0.9, 1.0)
    inp = th.cat((x, h), 1)
    v = F.linear(inp, p["f_val.weight"], p["f_val.bias"])
    # (6) W_sign = 0 and b_sign = 0 -> uniform attention -> c = mean of in-neighbour values
    p0 = dict(p, **{"f_sign.weight": th.zeros_like(p["f_sign.weight"]), "f_sign.bias": th.zeros_like(p["f_sign.bias"])})
    off, src = _talk([3, 4], "complete")
    g = dict(talk_off=off, talk_src=src)
    dst = R.seg_ids(off)
    c_mean = R.segment_mean(v[src.long()], dst, 7)
    ref = R.gru_cell(th.cat((x, c_mean), 1), h, R.sub(p0, "f_udt"))
    assert_close(R.tarmac(g, x, h, p0, key), ref, 1e-13)
    # single self loop -> c = v_self
    off, src = _talk([3, 4], "self")
    ref = R.gru_cell(th.cat((x, v), 1), h, R.sub(p, "f_udt"))
    assert_close(R.tarmac(dict(talk_off=off, talk_src=src), x, h, p, key), ref, 1e-13)


def test_dense_fp64_cross_check_talk_attention():
    H, msg, key = 16, 8, 4
    p = tarmac_params(H, msg, key)
    x, h = cf((7, H), 0.2, 1.0), cf((7, H), 0.9, 1.0)
    inp = th.cat((x, h), 1)
    v = F.linear(inp, p["f_val.weight"], p["f_val.bias"])
    s = F.linear(inp, p["f_sign.weight"], p["f_sign.bias"])
    q = F.linear(inp, p["f_que.weight"], p["f_que.bias"])
    for mode in ("complete", "ring", "self"):
        off, src = _talk([3, 4], mode)
        dst = R.seg_ids(off)
        e = (s[src.long()] * q[dst]).sum(-1, keepdim=True) / key
        c = R.segment_sum(v[src.long()] * R.segment_softmax(e, dst, 7), dst, 7)
        assert_close(c, Dn.talk_attention_dense(s, q, v, off, src, key), 1e-12, mode)


def test_kat_gru_cell_matches_aten_and_dueling_mean():
    H = 16
    p = tarmac_params(H)
    i, h = cf((5, H + 8), 0.4, 1.0), cf((5, H), 0.8, 1.0)
    ref = th._VF.gru_cell(i, h, p["f_udt.weight_ih"], p["f_udt.weight_hh"], p["f_udt.bias_ih"], p["f_udt.bias_hh"])
    assert_close(R.gru_cell(i, h, R.sub(p, "f_udt")), ref, 1e-13)
    pd = {"v_head.weight": cf((1, H), 1.1, 0.3), "v_head.bias": cf((1,), 1.2, 0.3),
          "adv_head.weight": cf((9, H), 1.3, 0.3), "adv_head.bias": cf((9,), 1.4, 0.3)}
    qd = R.q_head(h, pd, True)
    assert_close(qd.mean(-1, keepdim=True), F.linear(h, pd["v_head.weight"], pd["v_head.bias"]), 1e-13)  # (7)


def test_gradcheck_gatv2_and_tarmac():
    p = {k: v.clone().requires_grad_(True) for k, v in gat_params(d=4).items()}
    x_s, x_d, off = ragged([3, 0, 1, 5])
    keys = list(p)

    def f(*ws):
        return R.gatv2_conv_seg(x_s, x_d, off, dict(zip(keys, ws)), NH, activation=False)
    assert th.autograd.gradcheck(f, tuple(p.values()), eps=1e-6, atol=1e-7)
    H, msg, key = 8, 4, 2
    pt = {k: v.clone().requires_grad_(True) for k, v in tarmac_params(H, msg, key).items()}
    x = cf((5, H), 0.2, 1.0).requires_grad_(True)
    h = cf((5, H), 0.9, 1.0)   # h is stop-gradded on the message path (gnn_agents.py:254), so it is a constant here
    off, src = _talk([2, 3], "ring")
    kt = list(pt)

    def ft(x, *ws):
        return R.tarmac(dict(talk_off=off, talk_src=src), x, h, dict(zip(kt, ws)), key)
    assert th.autograd.gradcheck(ft, (x,) + tuple(pt.values()), eps=1e-6, atol=1e-7)


def test_disc_comm_literal_and_exact_tie_rules_pick_the_same_owner():
    """how often does the literal reference rule (max over (y_hard - y_soft) + y_soft in floating
    point, gnn_agents.py:166-178) route a channel's gradient to another in-edge than the exact rule K5 implements (first
    in-edge whose hard bit is set)?  Counted over complete 8-agent graphs (8 competing in-edges per destination) in
    float32 and float64: never - (1 - s) + s rounds to exactly 1.0 for every softmax output s drawn here, so both maxima
    are ties at 1.0 resolved towards the first edge."""
    import torch.nn.functional as F
    gen = th.Generator().manual_seed(0)
    for dt in (th.float32, th.float64):
        N, n, msg = 64, 8, 64
        src = (th.arange(N) // n * n).repeat_interleave(n) + th.arange(n).repeat(N)
        dst = th.arange(N).repeat_interleave(n)
        logits = th.randn(N, msg, 2, generator=gen).to(dt)[src]
        gum = -th.empty(N * n, msg, 2).exponential_(generator=gen).log().to(dt)
        y_soft = th.softmax((logits + gum) / 0.5, -1)
        y_hard = th.zeros_like(y_soft).scatter_(-1, y_soft.max(-1, keepdim=True)[1], 1.0)
        m = ((y_hard - y_soft) + y_soft).flatten(1)
        own_lit = R.segment_max_first(m, dst, N, return_owner=True)[1]
        own_exact = R.segment_max_first(m, dst, N, key=y_hard.flatten(1), return_owner=True)[1]
        assert int((own_lit != own_exact).sum()) == 0
        assert bool((m[y_hard.flatten(1) == 1] == 1).all())        # the rounding fact the equality rests on
