"""K3a + K3b in one launch (csrc/tarmac_msg.hip, include/uavgnn.h `uavgnn_tarmac_msg_*`) against the float64 oracle of the TarMAC
message (oracle/restatement.py `tarmac`, following gnn_agents.py:254-267): projections, attention weights, messages; every
supported graph size; ragged talk relations inside the graphs (edge-free graphs, parallel edges); row counts that are not a
multiple of the workgroup's 64; the operand planes handed to the GRU cell reproduce x, c and h EXACTLY (three-way bf16 split).
All calls go through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch as th

from oracle import restatement as R
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _uniform_talk(B, n, p, seed, dup=False):
    gen = th.Generator().manual_seed(seed)
    N = B * n
    src_l, dst_l = [], []
    for b in range(B):
        adj = th.rand(n, n, generator=gen) < (0.0 if b % 7 == 3 else p)
        i, j = adj.nonzero(as_tuple=True)
        src_l.append(i + b * n)
        dst_l.append(j + b * n)
    src, dst = th.cat(src_l), th.cat(dst_l)
    if dup and src.numel():
        keep = th.arange(src.numel())
        idx = th.sort(th.cat([keep, keep[::3]]))[0]
        src, dst = src[idx], dst[idx]
    o = th.argsort(dst * N + src, stable=True)
    src, dst = src[o], dst[o]
    off = th.zeros(N + 1, dtype=th.int32)
    off[1:] = th.cumsum(th.bincount(dst, minlength=N), 0)
    return N, off, src.to(th.int32), dst


def _planes_to_float(planes, N, H, M):
    """[x || c || h] back from the cell-operand planes: tiles [row block of 128][slice][plane][128 rows][4 chunks of 8 bf16], chunk g
    of row r at r * 4 + (g ^ ((r >> 2) & 3)); value = sum of the three planes (exact in float64)."""
    nsl = 2 * (H // 32) + (M + 31) // 32
    nrb = (N + 127) // 128
    raw = planes.cpu().numpy().view(np.uint16).reshape(nrb, nsl, 3, 128, 4, 8)
    f = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64).sum(2)      # [nrb, nsl, 128, 4 (position), 8]
    r = np.arange(128)
    out = np.empty((nrb, nsl, 128, 4, 8))
    for g in range(4):
        out[:, :, r, g, :] = f[:, :, r, g ^ ((r >> 2) & 3), :]
    return th.from_numpy(out.transpose(0, 2, 1, 3, 4).reshape(nrb * 128, nsl * 32)[:N])


@pytest.mark.parametrize("B,n,H,M,K,p,dup", [(37, 8, 256, 64, 16, 1.0, False), (13, 8, 256, 64, 16, 0.5, True),
                                             (9, 16, 64, 16, 8, 0.6, False), (50, 4, 64, 100, 14, 0.8, False),
                                             (129, 1, 32, 5, 3, 1.0, False), (33, 2, 96, 33, 7, 0.7, True),
                                             (4096, 8, 256, 64, 16, 1.0, False)])
def test_fused_tarmac_message_vs_oracle(B, n, H, M, K, p, dup):
    from uav_bs_ctrl_amd import _lib as L
    lib = L.lib()
    N, off, src, dst = _uniform_talk(B, n, p, seed=B + n, dup=dup)
    assert lib.uavgnn_tarmac_msg_supported(H, M, K, n) == 1
    gen = th.Generator().manual_seed(3)
    x, h = th.randn(N, H, generator=gen), th.randn(N, H, generator=gen)
    Wp, bp = th.randn(M + 2 * K, 2 * H, generator=gen) / (2 * H) ** 0.5, 0.1 * th.randn(M + 2 * K, generator=gen)
    # oracle, float64
    inp = th.cat((x, h), 1).double()
    proj64 = inp @ Wp.double().t() + bp.double()
    v64, s64, q64 = proj64[:, :M], proj64[:, M:M + K], proj64[:, M + K:]
    e = (s64[src.long()] * q64[dst]).sum(-1, keepdim=True) / K
    a64 = R.segment_softmax(e, dst, N)
    c64 = R.segment_sum(v64[src.long()] * a64, dst, N)

    dev = "cuda"
    xd, hd, Wd, bd, offd, srcd = (t.to(dev) for t in (x, h, Wp, bp, off, src))
    tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_tarmac_msg_prepare(Wd.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), L.stream()), "prepare")
    E = src.numel()

    def run(train, planes):
        ld_c = (H + M + 3) // 4 * 4 if train else M          # the x half is stored as float4s: row stride a multiple of 4
        inp_buf = th.full((N, ld_c), float("nan"), device=dev)
        a_save = th.full((max(E, 1),), float("nan"), device=dev) if train else None
        proj = th.full((N, M + 2 * K), float("nan"), device=dev) if train else None
        pl = (th.zeros(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M), dtype=th.uint8, device=dev) if planes else None)
        rc = lib.uavgnn_tarmac_msg_fwd(xd.data_ptr(), H, hd.data_ptr(), H, N, H, n, tiles.data_ptr(), bd.data_ptr(), M, K,
                                       offd.data_ptr(), L.ptr(srcd), 1.0 / K, inp_buf.data_ptr() + (4 * H if train else 0), ld_c,
                                       L.ptr(a_save), L.ptr(proj), M + 2 * K, inp_buf.data_ptr() if train else None, ld_c,
                                       L.ptr(pl), L.stream())
        L.check(rc, "uavgnn_tarmac_msg_fwd")
        th.cuda.synchronize()
        return inp_buf, a_save, proj, pl

    outs = {}
    for train in (False, True):
        for planes in (False, True):
            inp_buf, a_save, proj, pl = run(train, planes)
            c = inp_buf[:, H:H + M] if train else inp_buf
            assert_close(c, c64, 1e-5, f"c train={train} planes={planes}")
            if train:
                assert th.equal(inp_buf[:, :H], xd)
                assert_close(proj, proj64, 1e-5, "proj")
                if E:
                    assert_close(a_save[:E], a64[:, 0], 1e-5, "attention weights")
            if planes:
                rec = _planes_to_float(pl, N, H, M)
                want = th.cat((x.double(), c.cpu().double(), th.zeros(N, (-M) % 32, dtype=th.float64), h.double()), 1)
                assert th.equal(rec, want), "the operand planes are not an exact split of [x || c || h]"
            outs[(train, planes)] = c.clone()
    # one arithmetic for the training and the no-grad instantiation of a kernel, bit for bit, and run to run (the launches with
    # operand planes run on the one-wavefront-per-tile kernel - a single 2H-long chain -, the others on the wavefront-pair kernel -
    # (x part + bias) + h part: the two may differ in the last bit)
    for planes in (False, True):
        assert th.equal(outs[(True, planes)], outs[(False, planes)]), planes
    assert th.equal(run(False, False)[0], outs[(False, False)])
    assert th.equal(run(False, True)[0], outs[(False, True)])


def test_fused_tarmac_message_fails_loudly_on_edges_that_leave_their_graph():
    from uav_bs_ctrl_amd import _lib as L
    lib = L.lib()
    B, n, H, M, K = 10, 8, 64, 16, 8
    N, off, src, dst = _uniform_talk(B, n, 1.0, seed=1)
    src = src.clone()
    bad_graph = 4
    pos = int(off[bad_graph * n + 2])                    # first in-edge of an agent of graph 4 now comes from graph 0
    src[pos] = 1
    gen = th.Generator().manual_seed(0)
    x, h = th.randn(N, H, generator=gen).cuda(), th.randn(N, H, generator=gen).cuda()
    Wp, bp = (th.randn(M + 2 * K, 2 * H, generator=gen) / 16).cuda(), th.zeros(M + 2 * K).cuda()
    tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device="cuda")
    L.check(lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), L.stream()), "prepare")
    c = th.zeros(N, M, device="cuda")
    offd, srcd = off.cuda(), src.cuda()
    L.check(lib.uavgnn_tarmac_msg_fwd(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, offd.data_ptr(),
                                      srcd.data_ptr(), 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, None, L.stream()), "fwd")
    th.cuda.synchronize()
    nan_rows = th.isnan(c).any(1).view(B, n).all(1).cpu()              # the 16-row tile of the edge's destination: graphs 4 and 5
    assert bool(nan_rows[bad_graph]) and bool(nan_rows[bad_graph + 1]) and int(nan_rows.sum()) == 2
    assert not bool(th.isnan(c).view(B, n, M)[~nan_rows].any())
    # refusals: ragged row count, unsupported graph size, misaligned operand
    args = lambda N_, n_, xp: (xp, H, h.data_ptr(), H, N_, H, n_, tiles.data_ptr(), bp.data_ptr(), M, K, offd.data_ptr(),   # noqa: E731
                               srcd.data_ptr(), 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, None, L.stream())
    assert lib.uavgnn_tarmac_msg_fwd(*args(N - 1, n, x.data_ptr())) == L.UAVGNN_EUNSUPPORTED
    assert lib.uavgnn_tarmac_msg_fwd(*args(N, 5, x.data_ptr())) == L.UAVGNN_EUNSUPPORTED
    assert lib.uavgnn_tarmac_msg_fwd(*args(N, n, x.data_ptr() + 4)) == L.UAVGNN_EUNSUPPORTED
    assert lib.uavgnn_tarmac_msg_fwd(*args(N, n, None)) == L.UAVGNN_EINVAL


def test_tarmac_step_takes_the_fused_message_launch_and_equals_the_three_launch_path(monkeypatch):
    """ops.tarmac_step with and without csrc/tarmac_msg.hip (UAVGNN_MSG_FUSED): same outputs and gradients to fp32 rounding (the
    projection runs as bf16x3 products in one, as vendor fp32 GEMMs in the other), and the fused launch is actually taken."""
    import bench
    from uav_bs_ctrl_amd import ops
    from uav_bs_ctrl_amd.learner import MultiAgentQLearner
    taken = []
    orig = ops._launch_tarmac_msg

    def spy(*a, **k):
        taken.append(1)
        return orig(*a, **k)
    monkeypatch.setattr(ops, "_launch_tarmac_msg", spy)

    def grads(fused):
        monkeypatch.setattr(ops, "MSG_FUSED", fused)
        th.manual_seed(0)
        learner = MultiAgentQLearner(dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=3),
                                     bench.exp3_args("cuda"))
        batch = bench.make_sequence(160, 8, 20, 3, "env", th.device("cuda"), seed=7, distinct=2)
        learner.grads.zero_()
        out = learner.accumulate(batch)
        h = learner.init_hidden(160)
        acts, h2 = learner.act(batch["obs"][0], h, 0.0)
        return learner.grads.flat.clone(), float(out["LossQ"]), h2.clone()
    g1, l1, h1 = grads(True)
    n_taken = len(taken)
    g0, l0, h0 = grads(False)
    assert n_taken >= 7 + 1 and len(taken) == n_taken           # 2T + 1 update forwards + one act; none with the switch off
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    assert_close(h1, h0, 1e-5, "h' of act")
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("N,H,M", [(32768, 256, 64), (1000, 256, 64), (1280, 64, 32), (130, 128, 96)])
def test_gru_cell_from_operand_planes_is_bit_identical_to_the_in_kernel_split(N, H, M):
    """uavgnn_gru_cell_fwd_planes (csrc/gru_x3p.hip: LDS-DMA copies of the planes uavgnn_tarmac_msg_fwd wrote) against
    uavgnn_gru_cell_fwd_x3_cat (csrc/gru_x3.hip: the split inside the kernel) on the same [x || c], h: same products, same
    accumulation order - h' and the saved pre-activations equal bit for bit, with and without saves, ragged row counts included."""
    from uav_bs_ctrl_amd import _lib as L
    lib = L.lib()
    K, n = 16, 2
    dev = "cuda"
    gen = th.Generator().manual_seed(N)
    B = N // n
    _, off, src, _ = _uniform_talk(B, n, 1.0, seed=1)
    x, h = th.randn(N, H, generator=gen).to(dev), (0.5 * th.randn(N, H, generator=gen)).to(dev)
    Wp, bp = (th.randn(M + 2 * K, 2 * H, generator=gen) / (2 * H) ** 0.5).to(dev), th.zeros(M + 2 * K, device=dev)
    W_ih, W_hh = (th.randn(3 * H, H + M, generator=gen) / (H + M) ** 0.5).to(dev), (th.randn(3 * H, H, generator=gen) / H ** 0.5).to(dev)
    b_ih, b_hh = (0.1 * th.randn(3 * H, generator=gen)).to(dev), (0.1 * th.randn(3 * H, generator=gen)).to(dev)
    offd, srcd = off.to(dev), src.to(dev)
    tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), L.stream()), "prepare")
    c = th.empty(N, M, device=dev)
    planes = th.empty(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_tarmac_msg_fwd(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, offd.data_ptr(),
                                      srcd.data_ptr(), 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, planes.data_ptr(), L.stream()),
            "msg")
    w_planes = th.empty(lib.uavgnn_gru_cell_x3_workspace_bytes(H + M, H), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_gru_split_weights(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_planes.data_ptr(), L.stream()), "split")
    w_tiles = th.empty(lib.uavgnn_gru_weight_tiles_bytes(H + M, H), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_gru_split_weight_tiles(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_tiles.data_ptr(), L.stream()), "tiles")
    for save in (False, True):
        h_a, h_b = th.full((N, H), float("nan"), device=dev), th.full((N, H), float("nan"), device=dev)
        pre_a = th.full((N, 4 * H), float("nan"), device=dev) if save else None
        pre_b = th.full((N, 4 * H), float("nan"), device=dev) if save else None
        L.check(lib.uavgnn_gru_cell_fwd_x3_cat(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), N, H, w_planes.data_ptr(),
                                               b_ih.data_ptr(), b_hh.data_ptr(), h_a.data_ptr(), L.ptr(pre_a), L.stream()), "cell x3")
        L.check(lib.uavgnn_gru_cell_fwd_planes(planes.data_ptr(), H + M, h.data_ptr(), N, H, w_tiles.data_ptr(), b_ih.data_ptr(),
                                               b_hh.data_ptr(), h_b.data_ptr(), L.ptr(pre_b), L.stream()), "cell planes")
        th.cuda.synchronize()
        assert not bool(th.isnan(h_b).any())
        assert th.equal(h_a, h_b), f"h' differs: {float((h_a - h_b).abs().max())}"
        if save:
            assert th.equal(pre_a, pre_b)
        for opt in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13):          # every schedule of the kernel: the same bits
            h_c = th.full((N, H), float("nan"), device=dev)
            pre_c = th.full((N, 4 * H), float("nan"), device=dev) if save else None
            L.check(lib.uavgnn_gru_cell_fwd_planes_opts(planes.data_ptr(), H + M, h.data_ptr(), N, H, w_tiles.data_ptr(), b_ih.data_ptr(),
                                                        b_hh.data_ptr(), h_c.data_ptr(), L.ptr(pre_c), opt, L.stream()), f"opt {opt}")
            th.cuda.synchronize()
            assert th.equal(h_a, h_c), f"opt {opt}: h' differs"
            if save:
                assert th.equal(pre_a, pre_c), f"opt {opt}"
    # against the float64 oracle as well
    ref = R.gru_cell(th.cat((x, c), 1).double().cpu(), h.double().cpu(),
                     dict(weight_ih=W_ih.double().cpu(), weight_hh=W_hh.double().cpu(), bias_ih=b_ih.double().cpu(), bias_hh=b_hh.double().cpu()))
    assert_close(h_b, ref, 1e-5, "h' vs oracle")
