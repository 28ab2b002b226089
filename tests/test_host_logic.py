"""CPU tests of the host-side mirror: HeteroBatch container ops, state_dict layout, registry, loud failure."""
import types

import numpy as np
import pytest
import torch as th

from tests.util import load_golden
from uav_bs_ctrl_amd import REGISTRY, GnnAgent, HeteroBatch, batch, cat, from_obs_dicts, heterograph, merge
from uav_bs_ctrl_amd._lib import UavGnnError


def _args(c="tarmac", H=32, dueling=False):
    return types.SimpleNamespace(hidden_size=H, c=c, n_heads=4, n_layers=2, msg_size=8, key_size=4, n_rounds=1,
                                 dueling=dueling)


@pytest.mark.parametrize("name,c,duel", [("agent_tarmac", "tarmac", False), ("agent_none", None, False),
                                         ("agent_disc", "disc", False), ("agent_base", "base", False),
                                         ("agent_commnet", "commnet", False), ("agent_econv", "econv", False),
                                         ("agent_tarmac_duel", "tarmac", True)])
def test_state_dict_layout_matches_reference(name, c, duel):
    """Names, shapes and parameters() order equal the reference module's (captured in the golden fixtures)."""
    _, _, p, cfg, z = load_golden(name)
    net = GnnAgent(dict(agent=2, ubs=2, gt=4), cfg["n_actions"], _args(c, dueling=duel))
    names = [k for k, _ in net.named_parameters()]
    assert names == [str(n) for n in z["param_names"]]
    for k, prm in net.named_parameters():
        assert tuple(prm.shape) == tuple(p[k].shape), k
    net.load_state_dict(p)   # strict


def test_registry_and_errors():
    assert REGISTRY["gnn"] is GnnAgent
    with pytest.raises(KeyError):
        GnnAgent(dict(agent=2, ubs=2, gt=4), 9, _args("nope"))
    with pytest.raises(AssertionError):
        GnnAgent(dict(agent=2, ubs=2, gt=4), 9, types.SimpleNamespace(hidden_size=30, c=None, n_heads=4, n_layers=1,
                                                                       msg_size=8, key_size=4, n_rounds=1, dueling=False))
    net = GnnAgent(dict(agent=2, ubs=2, gt=4), 9, _args())
    assert net.init_hidden().shape == (1, 32) and net.init_hidden().device.type == "cpu"


def test_missing_res_fc_bias_is_tolerated():
    _, _, p, cfg, _ = load_golden("agent_tarmac")
    p = {k: v for k, v in p.items() if not k.endswith("res_fc.bias")}
    net = GnnAgent(dict(agent=2, ubs=2, gt=4), 9, _args())
    net.load_state_dict(p)
    assert float(net.enc.f_conv["seen"].res_fc.bias.abs().max()) == 0.0


def test_no_cpu_fallback():
    g, h, p, cfg, _ = load_golden("agent_tarmac", dtype=th.float32)
    net = GnnAgent(dict(agent=2, ubs=2, gt=4), 9, _args())
    with pytest.raises(UavGnnError):
        net(HeteroBatch.from_arrays(**g), h)


def _obs(rng, n, M):
    out = []
    for i in range(n):
        gt = rng.uniform(-1, 1, (M, 5)).astype(np.float32)
        gt[:, 0] = rng.uniform(size=M) < 0.5
        ub = rng.uniform(-1, 1, (n - 1, 3)).astype(np.float32)
        ub[:, 0] = rng.uniform(size=n - 1) < 0.6
        out.append(dict(agent=rng.uniform(0, 1, 2).astype(np.float32), ubs=ub, gt=gt))
    return out


def test_from_obs_dicts_equals_reference_style_construction():
    """Vectorised builder == per-agent heterograph + batch + merge (the reference's env_wrappers.py:65-89,:122-154)."""
    rng = np.random.default_rng(0)
    n, M = 5, 12
    obs = _obs(rng, n, M)
    d = rng.uniform(0, 2, (n, n))
    d = (d + d.T) / 2
    np.fill_diagonal(d, 0)
    fast = from_obs_dicts(obs, d, r_comm=1.0)
    # slow path, spelled like the reference
    locs = []
    for o in obs:
        gi, ui = o["gt"][:, 0] == 1, o["ubs"][:, 0] == 1
        g = heterograph({("gt", "seen", "agent"): (np.arange(gi.sum()), np.zeros(gi.sum(), dtype=np.int64)),
                         ("ubs", "near", "agent"): (np.arange(ui.sum()), np.zeros(ui.sum(), dtype=np.int64)),
                         ("agent", "talk", "agent"): ([], [])},
                        num_nodes_dict={"gt": gi.sum(), "ubs": ui.sum(), "agent": 1})
        g.ndata["feat"] = {"gt": th.as_tensor(o["gt"][gi, 1:]), "ubs": th.as_tensor(o["ubs"][ui, 1:]),
                           "agent": th.as_tensor(o["agent"]).unsqueeze(0)}
        locs.append(g)
    u, v = [], []
    for i in range(n):
        for j in range(n):
            if d[i, j] <= 1.0:
                u.append(i), v.append(j)
    comm = heterograph({("gt", "seen", "agent"): ([], []), ("ubs", "near", "agent"): ([], []),
                        ("agent", "talk", "agent"): (u, v)}, num_nodes_dict={"gt": 0, "ubs": 0, "agent": n})
    slow = merge([batch(locs), comm])
    for et in ("seen", "near"):
        xs, offs = slow.relation_segments(et)
        xf, offf = fast.relation_segments(et)
        assert th.equal(offs, offf) and th.equal(xs, xf)
    assert all(th.equal(a, b) for a, b in zip(slow.talk_csc(), fast.talk_csc()))
    assert th.equal(slow.talk_eid(), fast.talk_eid())
    assert th.equal(slow.agent_feat(), fast.agent_feat())
    # transpose is consistent: position t_pos[k] of the CSC holds an edge whose source is the segment owner
    t_off, t_dst, t_pos = fast.talk_transpose()
    off, src = fast.talk_csc()
    owner = th.repeat_interleave(th.arange(n), (t_off[1:] - t_off[:-1]).long())
    assert th.equal(src[t_pos.long()].long(), owner)
    from uav_bs_ctrl_amd.graph import seg_ids
    assert th.equal(seg_ids(off)[t_pos.long()], t_dst.long())


def test_batch_offsets_and_cat():
    rng = np.random.default_rng(1)
    gs = []
    for k in range(3):
        n = 3 + k
        d = np.zeros((n, n))
        gs.append(from_obs_dicts(_obs(rng, n, 7), d, r_comm=1.0))
    b = cat(gs)
    assert b.num_nodes("agent") == 3 + 4 + 5
    assert b.graph_off.tolist() == [0, 3, 7, 12]
    off, src = b.talk_csc()
    assert int(off[-1]) == 9 + 16 + 25 and int(src.max()) == 11
    # talk edges never cross environments
    from uav_bs_ctrl_amd.graph import seg_ids
    dst = seg_ids(off)
    env_of = th.repeat_interleave(th.arange(3), th.tensor([3, 4, 5]))
    assert th.equal(env_of[src.long()], env_of[dst])
    xs, so = b.relation_segments("seen")
    assert xs.shape[0] == int(so[-1]) and so.numel() == 13
    assert th.equal(cat([th.ones(2, 3), th.zeros(1, 3)]), th.cat([th.ones(2, 3), th.zeros(1, 3)]))
    with pytest.raises(TypeError):
        cat([1, 2])


def test_general_unsorted_relation_is_gathered_into_segments():
    g = heterograph({("gt", "seen", "agent"): ([2, 0, 1, 3], [1, 0, 1, 0]), ("ubs", "near", "agent"): ([], []),
                     ("agent", "talk", "agent"): ([0, 1], [1, 0])}, num_nodes_dict={"gt": 4, "ubs": 0, "agent": 2})
    feat = th.arange(16, dtype=th.float32).view(4, 4)
    g.ndata["feat"] = {"gt": feat, "agent": th.zeros(2, 2), "ubs": th.zeros(0, 2)}
    xs, off = g.relation_segments("seen")
    assert off.tolist() == [0, 2, 4]
    assert th.equal(xs, feat[[0, 3, 2, 1]])


def test_slice_agents_views_and_relation_order():
    rng = np.random.default_rng(4)
    gs = [from_obs_dicts(_obs(rng, 4, 29), np.zeros((4, 4)), r_comm=1.0) for _ in range(3)]
    big = batch(gs)
    assert big.hints == {"max_graph_agents": 4, "max_deg:seen": 29, "max_deg:near": 3}
    assert big.relation_order("near") is None          # bounded by one 16-edge row tile: nothing to balance
    mid = big.slice_agents(4, 8)                       # the agents of the second environment
    for et in ("seen", "near"):
        xs, off = mid.relation_segments(et)
        xr, offr = gs[1].relation_segments(et)
        assert th.equal(off, offr) and th.equal(xs, xr)
        assert xs.data_ptr() >= big.relation_segments(et)[0].data_ptr()      # a view, not a copy
    assert th.equal(mid.agent_feat(), gs[1].agent_feat()) and not mid.has_relation("talk")
    assert big.relation_order("seen") is None          # fewer destinations than persistent wavefronts: no order needed
    N = 3000
    deg = th.as_tensor(rng.integers(0, 40, N))
    deg[th.as_tensor(rng.random(N) < 0.5)] = 0         # sparse: mean in-degree below 16, isolated destinations to skip
    off = th.zeros(N + 1, dtype=th.int32)
    off[1:] = th.cumsum(deg, 0)
    wide = HeteroBatch.from_arrays(x_a=th.zeros(N, 2), x_gt=th.zeros(int(off[-1]), 4), seen_off=off)
    order = wide.relation_order("seen")
    d = deg[order.long()]
    assert sorted(order.tolist()) == list(range(N)) and bool((d[1:] <= d[:-1]).all())
    # dense relations (mean in-degree >= 16) are handed out in natural order: the round robin over the persistent wavefronts
    # balances by itself, the order's indirection and its four launches only cost (graph.py: relation_order)
    deg = th.as_tensor(rng.integers(20, 40, N))
    off = th.zeros(N + 1, dtype=th.int32)
    off[1:] = th.cumsum(deg, 0)
    dense = HeteroBatch.from_arrays(x_a=th.zeros(N, 2), x_gt=th.zeros(int(off[-1]), 4), seen_off=off)
    assert dense.relation_order("seen") is None
    assert big.graph_off.tolist() == [0, 4, 8, 12]


def test_tuned_gemm_selection_is_inert_without_a_gpu():
    """The recorded vendor-GEMM solutions ship with the package; selecting them is a no-op on a CPU-only host."""
    import os
    from uav_bs_ctrl_amd import tuned
    assert os.path.exists(tuned.CSV)
    head = open(tuned.CSV).read().splitlines()
    assert head[0].startswith("Validator,PT_VERSION") and any(l.startswith("Gemm") for l in head)
    if not th.cuda.is_available():
        assert tuned.enable_tuned_gemms() is False


def test_time_split_gradient_buffer_matches_unbind():
    """ops.time_split: per-step views whose gradients land in one buffer.  A consumer that writes into its slot
    (as the fused step does) and one that returns a fresh tensor (any other step) must both give unbind's gradient."""
    from uav_bs_ctrl_amd import ops

    class _WritesIntoSlot(th.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, slot):
            ctx.save_for_backward(w)
            ctx.slot = slot
            return x * w

        @staticmethod
        def backward(ctx, g):
            (w,) = ctx.saved_tensors
            th.mul(g, w, out=ctx.slot)
            return ctx.slot, None, None

    T1, N, H = 5, 7, 3
    gen = th.Generator().manual_seed(0)
    x_all = th.randn(T1 * N, H, generator=gen, requires_grad=True)
    w = th.randn(T1, N, H, generator=gen)
    ref = th.autograd.grad(sum(((xt * w[t]) ** 2).sum() for t, xt in enumerate(x_all.view(T1, N, H).unbind(0))), x_all)[0]
    xs, slots = ops.time_split(x_all, T1)
    assert all(s is not None for s in slots)
    terms = []
    for t in range(T1):
        y = _WritesIntoSlot.apply(xs[t], w[t], slots[t]) if t % 2 == 0 else xs[t] * w[t]     # slot writer / stray
        terms.append((y ** 2).sum())
    got = th.autograd.grad(sum(terms[:-1]), x_all)[0]        # the last step is unused: its slice must come back zero
    ref2 = ref.clone().view(T1, N, H)
    ref2[-1] = 0
    assert th.allclose(got, ref2.view(-1, H), rtol=1e-6, atol=1e-7)
    with th.no_grad():
        xs2, slots2 = ops.time_split(x_all, T1)
    assert slots2 == [None] * T1 and th.equal(xs2[2], x_all.view(T1, N, H)[2])


def test_row_blocked_column_sums():
    from uav_bs_ctrl_amd import ops
    assert ops._row_blocks(32768) == 256 and ops._row_blocks(96) == 2 and ops._row_blocks(7) == 1
    x = th.randn(640, 11, generator=th.Generator().manual_seed(1))
    assert th.allclose(ops._colsum(x), x.sum(0), rtol=1e-5, atol=1e-5)
    assert th.allclose(ops._colsum(x[:, 3:9]), x[:, 3:9].sum(0), rtol=1e-5, atol=1e-5)


def _reference_style_env_graph(obs, d, r_comm):
    """The reference's construction spelled with this package's dgl-like calls (env_wrappers.py:65-89,:122-154)."""
    n = len(obs)
    locs = []
    for o in obs:
        gi, ui = o["gt"][:, 0] == 1, o["ubs"][:, 0] == 1
        g = heterograph({("gt", "seen", "agent"): (np.arange(gi.sum()), np.zeros(gi.sum(), dtype=np.int64)),
                         ("ubs", "near", "agent"): (np.arange(ui.sum()), np.zeros(ui.sum(), dtype=np.int64)),
                         ("agent", "talk", "agent"): ([], [])},
                        num_nodes_dict={"gt": gi.sum(), "ubs": ui.sum(), "agent": 1})
        g.ndata["feat"] = {"gt": th.as_tensor(o["gt"][gi, 1:]), "ubs": th.as_tensor(o["ubs"][ui, 1:]),
                           "agent": th.as_tensor(o["agent"]).unsqueeze(0)}
        locs.append(g)
    u, v = [], []
    for i in range(n):
        for j in range(n):
            if d[i, j] <= r_comm:
                u.append(i), v.append(j)
    comm = heterograph({("gt", "seen", "agent"): ([], []), ("ubs", "near", "agent"): ([], []),
                        ("agent", "talk", "agent"): (u, v)}, num_nodes_dict={"gt": 0, "ubs": 0, "agent": n})
    return merge([batch(locs), comm])


def test_merge_takes_graph_boundaries_from_the_talk_holder():
    """dgl.merge([dgl.batch(per-agent graphs), comm_graph]) must yield ONE graph of n agents - the
    per-agent boundaries [0,1,..,n] of the observation operand do not delimit the talk relation."""
    rng = np.random.default_rng(4)
    n = 5
    g = _reference_style_env_graph(_obs(rng, n, 9), np.zeros((n, n)), 1.0)
    assert g.graph_off.tolist() == [0, n] and g.hints["max_graph_agents"] == n
    fast = from_obs_dicts(_obs(np.random.default_rng(4), n, 9), np.zeros((n, n)), 1.0)
    assert fast.graph_off.tolist() == g.graph_off.tolist() and fast.hints["max_graph_agents"] == n
    b = batch([g, g, g])
    assert b.graph_off.tolist() == [0, n, 2 * n, 3 * n] and b.hints["max_graph_agents"] == n
    # every talk source lies inside the graph of its destination
    off, src = b.talk_csc()
    dst = th.repeat_interleave(th.arange(3 * n), (off[1:] - off[:-1]).long())
    assert th.equal(src.long() // n, dst // n)


def test_fused_projection_follows_data_inplace_polyak():
    """the reference's polyak update is ``p_targ.data.mul_() / .data.add_()`` (learner.py:165-166),
    which bumps no version counter; the stacked projection weight of the fused TarMAC step must see it."""
    pol = GnnAgent(dict(agent=2, ubs=2, gt=4), 9, _args())
    tgt = GnnAgent(dict(agent=2, ubs=2, gt=4), 9, _args())
    tgt.load_state_dict(pol.state_dict())
    W0 = tgt.f_comm.fused_projection()[0].clone()
    with th.no_grad():
        for p in pol.parameters():
            p.add_(1.0)
    for p, p_targ in zip(pol.parameters(), tgt.parameters()):          # the reference's loop, verbatim
        p_targ.data.mul_(0.9)
        p_targ.data.add_((1 - 0.9) * p.data)
    W1, b1 = tgt.f_comm.fused_projection()
    c = tgt.f_comm
    assert th.equal(W1, th.cat((c.f_val.weight, c.f_sign.weight, c.f_que.weight), 0))
    assert th.equal(b1, th.cat((c.f_val.bias, c.f_sign.bias, c.f_que.bias), 0))
    assert float((W1 - (W0 + 0.1)).abs().max()) < 1e-6
    c.f_que.weight.data = c.f_que.weight.data.clone() * 2.0             # storage REPLACED: caught by the pointer check
    assert th.equal(c.fused_projection()[0][-c.f_que.weight.shape[0]:], c.f_que.weight)
    sd = {k: v.clone() for k, v in pol.state_dict().items()}
    tgt.load_state_dict(sd)
    assert th.equal(tgt.f_comm.fused_projection()[0][:c.f_val.weight.shape[0]], pol.f_comm.f_val.weight)


def test_weight_grad_sink_keeps_partials_when_the_chunk_count_changes():
    """a change of N (hence of the row-chunk count S) between two accumulations must not drop the
    gradient accumulated so far."""
    from uav_bs_ctrl_amd.ops import WeightGradSink
    gen = th.Generator().manual_seed(0)
    sink, got = WeightGradSink(), []
    parts = []
    for n in (4096, 8192, 4096, 100):
        dy, x = th.randn(n, 6, generator=gen), th.randn(n, 5, generator=gen)
        parts.append(dy.t() @ x)
        sink.weight("W", dy, x, lambda g: got.append(g))
    assert len({k[1] for k in sink.slots}) == 3
    sink.flush()
    assert th.allclose(sum(got), sum(parts), rtol=1e-4, atol=1e-3)


def test_checkpoint_carries_the_lr_scheduler_like_the_reference(tmp_path):
    """learner.py:50-56,:183-184,:198-199: anneal_lr (default True in madrqn/config.py:36) adds a LambdaLR
    max(0.4, 1 - epoch/100) and its state to the checkpoint."""
    from uav_bs_ctrl_amd.learner import MultiAgentQLearner
    args = types.SimpleNamespace(device="cpu", hidden_size=32, c="tarmac", n_heads=4, n_layers=2, msg_size=8, key_size=4,
                                 n_rounds=1, dueling=False, mixer=False, double_q=True, lr=5e-4, gamma=0.99,
                                 polyak=0.999, max_seq_len=5, seed=0, anneal_lr=True)
    info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=3, episode_limit=5)
    L = MultiAgentQLearner(info, args)
    for _ in range(70):
        L.lr_scheduler.step()
    assert abs(L.optimizer.param_groups[0]["lr"] - 5e-4 * 0.4) < 1e-12       # floor of the schedule
    path = str(tmp_path / "ck.pt")
    L.save_checkpoint(path, dict(epoch=70, t=123))
    ck = th.load(path, weights_only=False)
    assert {"epoch", "t", "model_state_dict", "optimizer_state_dict", "lr_scheduler_state_dict"} <= set(ck)
    L2 = MultiAgentQLearner(info, args)
    assert L2.load_checkpoint(path) == dict(epoch=70, t=123)
    assert L2.lr_scheduler.last_epoch == 70
    for (k, a), (_, b) in zip(L.policy_net.state_dict().items(), L2.policy_net.state_dict().items()):
        assert th.equal(a, b), k


def test_heterobatch_to_same_device_keeps_the_object():
    g = from_obs_dicts(_obs(np.random.default_rng(0), 3, 5), np.zeros((3, 3)), 1.0)
    assert g.to("cpu") is g and g.to(th.device("cpu")) is g


def test_weight_plane_cache_lives_only_inside_a_frozen_weights_scope():
    """ops._cached_planes (host logic of the bf16x3 wrappers): outside ops.frozen_weights() every call builds its planes again
    (a drop-in module's weights may change behind any cache); inside, once per key, nested scopes share the outer cache and
    the cache is gone when the outermost scope exits - also when it exits by an exception."""
    from uav_bs_ctrl_amd import ops
    built = []
    build = lambda p: built.append(p.numel())  # noqa: E731
    ops._cached_planes(("k", 1), 16, "cpu", build)
    ops._cached_planes(("k", 1), 16, "cpu", build)
    assert built == [16, 16] and ops._PLANES is None
    with ops.frozen_weights():
        a = ops._cached_planes(("k", 1), 16, "cpu", build)
        b = ops._cached_planes(("k", 1), 16, "cpu", build)
        with ops.frozen_weights():
            c = ops._cached_planes(("k", 1), 16, "cpu", build)
            ops._cached_planes(("k", 2), 32, "cpu", build)
        assert a is b and b is c and built == [16, 16, 16, 32] and len(ops._PLANES) == 2
    assert ops._PLANES is None
    try:
        with ops.frozen_weights():
            ops._cached_planes(("k", 3), 8, "cpu", build)
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert ops._PLANES is None and built[-1] == 8


def test_kernel_timer_filter_selects_spans_by_name():
    """ops.KERNEL_TIMER.reset(only=...) (bench.py times only the graded kernel inside its timed region): a span outside the
    filter, or any span of a disabled timer, must not touch the HIP event API at all."""
    from uav_bs_ctrl_amd import ops
    t = ops._KernelTimer()
    t.reset(enabled=True, only=("graded",))
    with t.span("other") as s:
        assert s.on is False                 # no event was created (th.cuda.Event would fail on this CPU-only box anyway)
    t.reset(enabled=False)
    with t.span("graded") as s:
        assert s.on is False
    assert t._spans == {}


def test_weight_plane_cache_keeps_the_weights_it_is_keyed_on_alive():
    """The cache key holds weight ADDRESSES: the entry holds the weights, so the allocator cannot recycle an address while a
    scope could still hit it (TarMAC's per-call th.cat weight of the target net vs the policy net's next one)."""
    import weakref
    from uav_bs_ctrl_amd import ops
    w = th.zeros(8)
    ref = weakref.ref(w)
    with ops.frozen_weights():
        ops._cached_planes(("mat", w.data_ptr()), 8, "cpu", lambda p: None, keep=(w,))
        del w
        assert ref() is not None
    assert ref() is None


def test_frozen_weights_store_outlives_the_scope_only_when_the_caller_owns_it():
    """ops.frozen_weights(store): the learner's rollout store (MultiAgentQLearner.act) - entries survive the scope and are reused
    by the next scope over the same dict until the owner clears it; a nested default scope shares it; a store that is fed with
    never-repeating keys (weights built by th.cat per call) is bounded."""
    from uav_bs_ctrl_amd import ops
    built = []
    build = lambda p: built.append(p.numel())  # noqa: E731
    store = {}
    with ops.frozen_weights(store):
        a = ops._cached_planes(("w", 1), 16, "cpu", build)
        with ops.frozen_weights():
            assert ops._cached_planes(("w", 1), 16, "cpu", build) is a
    assert ops._PLANES is None and len(store) == 1
    with ops.frozen_weights(store):
        assert ops._cached_planes(("w", 1), 16, "cpu", build) is a and built == [16]
    store.clear()                                        # what learner.invalidate_weight_cache() does
    with ops.frozen_weights(store):
        assert ops._cached_planes(("w", 1), 16, "cpu", build) is not a and built == [16, 16]
        for i in range(3 * ops._PLANES_MAX):
            ops._cached_planes(("tmp", i), 4, "cpu", build)
        assert len(store) <= ops._PLANES_MAX


def test_learner_drops_its_rollout_cache_wherever_it_moves_the_parameters(tmp_path):
    """MultiAgentQLearner keeps what `act` derived from the policy's parameters until apply / load_checkpoint /
    invalidate_weight_cache (CPU harness: the store is exercised with a marker entry; the HIP path fills it itself)."""
    import test_dp_gloo as dp
    import uav_bs_ctrl_amd.learner as LM
    saved = LM.agent_REGISTRY
    try:
        learner, batch = dp._learner(), dp._make_batch(0, 4)      # (the helper swaps the agent registry for its CPU stand-in)
    finally:
        LM.agent_REGISTRY = saved
    learner._rollout_planes[("marker",)] = (th.zeros(1), ())
    learner.update(batch)                                # accumulate -> apply
    assert len(learner._rollout_planes) == 0
    learner._rollout_planes[("marker",)] = (th.zeros(1), ())
    path = str(tmp_path / "ck.pt")
    learner.save_checkpoint(path, dict(epoch=0, t=0))
    assert len(learner._rollout_planes) == 1             # saving moves nothing
    learner.load_checkpoint(path)
    assert len(learner._rollout_planes) == 0
    learner._rollout_planes[("marker",)] = (th.zeros(1), ())
    learner.invalidate_weight_cache()
    assert len(learner._rollout_planes) == 0
