"""CPU tests of the tensor-native replay (buffer.py semantics) and of the QMixer against the reference fixture."""
import math
import types

import numpy as np
import pytest
import torch as th

from oracle.closed_form import closed_form_tensor
from tests.util import GOLDEN, assert_close
from uav_bs_ctrl_amd.agents.qmix import QMixer
from uav_bs_ctrl_amd.replay import SequenceReplay


def _transition(rng, E, n, M, H, t):
    f = lambda *s: th.as_tensor(rng.uniform(-1, 1, s).astype(np.float32))  # noqa: E731
    gt, ub = f(E, n, M, 5), f(E, n, n - 1, 3)
    gt[..., 0] = (gt[..., 0] > 0).float()
    ub[..., 0] = (ub[..., 0] > 0).float()
    return dict(gt=gt, ubs=ub, agent=f(E, n, 2).abs(), d_u2u=f(E, n, n).abs(), h=f(E, n, H), state=f(E, 3),
                act=th.full((E, n), t), rew=f(E, n), done=th.zeros(E, 1))


def test_sequence_semantics_match_reference_buffer():
    """T pushes make one sequence per env holding T transitions + the next obs/h of the last one (buffer.py:26-35)."""
    rng = np.random.default_rng(0)
    E, n, M, H, T = 2, 3, 5, 4, 3
    rb = SequenceReplay(capacity=4, max_seq_len=T, n_agents=n, n_gts=M, hidden_size=H, n_envs=E, state_dim=3,
                        r_comm=0.5, device="cpu")
    trs = [_transition(rng, E, n, M, H, t) for t in range(2 * T + 1)]
    for t in range(2 * T):
        tr = dict(trs[t])
        for k in ("gt", "ubs", "agent", "d_u2u", "h", "state"):
            tr["next_" + k] = trs[t + 1][k]
        rb.push(tr)
        assert len(rb) == (E if t + 1 >= T else 0) + (E if t + 1 >= 2 * T else 0)
    assert len(rb) == 4 and rb.ptr == 0
    b = rb.gather(th.tensor([0, 1, 2, 3]))
    assert len(b["obs"]) == T + 1 and b["obs"][0].num_nodes("agent") == 4 * n
    assert b["acts"].shape == (T, 4 * n, 1) and b["rews"].shape == (T, 4, n) and b["dones"].shape == (T, 4, 1)
    assert th.equal(b["acts"][:, 0, 0], th.arange(T)) and th.equal(b["acts"][:, 2 * n, 0], th.arange(T, 2 * T))
    # sequence 0 (env 0, first T steps): h0/h1 are the stored hidden states of its first two steps
    assert th.equal(b["h0"][:n], trs[0]["h"][0]) and th.equal(b["h1"][:n], trs[1]["h"][0])
    # the (T+1)-th observation of the first sequence is the next observation of its last transition
    assert th.equal(b["obs"][T].agent_feat()[:n], trs[T]["agent"][0])
    # graphs are rebuilt exactly as the per-step builder would: visible rows only, reference order
    xs, off = b["obs"][1].relation_segments("seen")
    keep = trs[1]["gt"][0, 0, :, 0] == 1
    assert int(off[1]) == int(keep.sum()) and th.equal(xs[:int(off[1])], trs[1]["gt"][0, 0][keep][:, 1:])
    # ring wrap-around overwrites the oldest sequences
    for t in range(T):
        rb.push(dict(trs[t], **{"next_" + k: trs[t + 1][k] for k in ("gt", "ubs", "agent", "d_u2u", "h", "state")}))
    assert len(rb) == 4 and rb.head == 2
    idx = rb.sample_indices(3, th.Generator().manual_seed(0))
    assert idx.unique().numel() == 3


def test_qmixer_matches_reference_fixture():
    z = np.load(f"{GOLDEN}/qmixer.npz")
    T, B, n = z["qs"].shape
    S = z["states"].shape[-1]
    mix = QMixer(S, n, types.SimpleNamespace(embed_dim=8)).double()
    names = [k for k, _ in mix.named_parameters()]
    assert names == [str(k) for k in z["param_names"]]
    with th.no_grad():
        for i, (k, p) in enumerate(mix.named_parameters()):
            assert repr(tuple(p.shape)) == str(z["param_shapes"][i])
            p.copy_(closed_form_tensor(p.shape, 1.0 + i * math.pi / 7, 0.1 if p.dim() == 1 else 0.25, th.float64))
    qs = th.as_tensor(z["qs"]).requires_grad_(True)
    y = mix(qs, th.as_tensor(z["states"]))
    assert_close(y, th.as_tensor(z["y"]), 1e-12, "q_tot")
    grads = th.autograd.grad((y * th.as_tensor(z["w"])).sum(), list(mix.parameters()) + [qs])
    for k, g in zip(names + ["__qs__"], grads):
        assert_close(g, th.as_tensor(z["grad:" + k]), 1e-10, f"grad {k}", floor=1e-14)


def test_sequence_replay_reproduces_the_reference_replay_buffer():
    """Row f2 against the reference's own ReplayBuffer (buffer.py:7-42) as filled by its learner.cache
    (learner.py:82-92) in a rollout crossing an episode end (tests/golden/replay_buffer.npz, make_golden.py `replay`):
    the same pushed transitions must leave the same sequences - cut every T pushes whatever the episode does, T+1
    observations / hidden states / states per sequence, next_h zeroed at the episode end, reference graphs per step.
    Host ring + host graph builder here; the `-m gpu` twin below runs the device ring + the HIP builder."""
    _check_replay_against_reference_buffer("cpu")


@pytest.mark.gpu
def test_sequence_replay_on_the_device_reproduces_the_reference_replay_buffer():
    """Row f2 on the GPU: ring in HBM -> index gather -> HIP graph builder (csrc/build_graph.hip), bit-exact against
    what the reference's ReplayBuffer + env wrapper produced from the same pushes."""
    _check_replay_against_reference_buffer("cuda")


def test_learner_cache_applies_the_reference_rules_to_the_raw_transition():
    """MultiAgentQLearner.cache (learner.py:82-92): fed the RAW arguments the reference's cache received in the captured
    rollout - done and bad_mask as the env returned them, next_h before it is zeroed - it must leave the sequences the
    reference's cache + ReplayBuffer left: done muted by bad_mask (push 9: the episode ends by its time limit), next_h zeroed
    by the raw done flag.  share_reward: the team mean, stored once."""
    _check_replay_against_reference_buffer("cpu", via_cache=True)
    import types
    from uav_bs_ctrl_amd.learner import MultiAgentQLearner
    rb = SequenceReplay(capacity=4, max_seq_len=2, n_agents=3, n_gts=2, hidden_size=4, n_envs=2, rew_dim=1, device="cpu")
    me = types.SimpleNamespace(n_agents=3, args=types.SimpleNamespace(share_reward=True))
    o = dict(gt=th.zeros(2, 3, 2, 5), ubs=th.zeros(2, 3, 2, 3), agent=th.zeros(2, 3, 2), d_u2u=th.zeros(2, 3, 3))
    rew = th.tensor([[1.0, 2.0, 6.0], [0.0, 0.0, 3.0]])
    MultiAgentQLearner.cache(me, rb, o, th.ones(6, 4), None, th.zeros(6, dtype=th.long), rew, o, th.ones(6, 4), None,
                             th.tensor([1.0, 0.0]), th.tensor([0.0, 0.0]))
    assert rb.cur["rew"][:, 0].flatten().tolist() == [3.0, 1.0]           # team means
    assert rb.cur["done"][:, 0].flatten().tolist() == [1.0, 0.0]          # a real terminal state stays done


@pytest.mark.gpu
def test_learner_cache_on_the_device_applies_the_reference_rules():
    _check_replay_against_reference_buffer("cuda", via_cache=True)


def _check_replay_against_reference_buffer(device, via_cache=False):
    import ast
    z = np.load(f"{GOLDEN}/replay_buffer.npz")
    meta = ast.literal_eval(str(z["meta"]))
    T, H, n, M = meta["T"], meta["H"], meta["n_agents"], meta["n_gts"]
    rb = SequenceReplay(capacity=10, max_seq_len=T, n_agents=n, n_gts=M, hidden_size=H, n_envs=1,
                        state_dim=meta["state_dim"], r_comm=float(z["r_comm"]), rew_dim=n, device=device)
    f = lambda k: th.as_tensor(z[k]).to(device)  # noqa: E731
    for i in range(meta["n_pushes"]):
        tr = {}
        for k in ("gt", "ubs", "agent", "d_u2u", "h", "state"):
            tr[k] = f(f"push{i}:{k}").unsqueeze(0) if k != "state" else f(f"push{i}:{k}").reshape(1, -1)
            nk = f(f"push{i}:next_{k}")
            tr["next_" + k] = nk.unsqueeze(0) if k != "state" else nk.reshape(1, -1)
        tr["act"] = f(f"push{i}:act").reshape(1, n)
        tr["rew"] = f(f"push{i}:rew").reshape(1, n)
        tr["done"] = f(f"push{i}:done").reshape(1, 1)
        if via_cache:     # the raw arguments through the learner's cache (the reference's run had share_reward = False)
            import types
            from uav_bs_ctrl_amd.learner import MultiAgentQLearner
            me = types.SimpleNamespace(n_agents=n, args=types.SimpleNamespace(share_reward=False))
            obs = {k: tr[k] for k in ("gt", "ubs", "agent", "d_u2u")}
            nxt = {k: tr["next_" + k] for k in ("gt", "ubs", "agent", "d_u2u")}
            MultiAgentQLearner.cache(me, rb, obs, tr["h"], tr["state"], f(f"push{i}:act"), f(f"push{i}:raw_rew"), nxt,
                                     f(f"push{i}:raw_next_h"), tr["next_state"], float(z[f"push{i}:raw_done"]),
                                     float(z[f"push{i}:raw_bad_mask"]))
        else:
            rb.push(tr)
    assert len(rb) == meta["n_seqs"] == 4 and rb.ptr == meta["n_pushes"] - 4 * T
    b = rb.gather(th.arange(4, device=device))
    if device == "cuda":
        assert all(g.agent_feat().is_cuda for g in b["obs"])
    for si in range(4):
        for t in range(T + 1):
            g = b["obs"][t]
            lo, hi = si * n, (si + 1) * n
            ref = {k.split(":")[-1]: z[k] for k in z.files if k.startswith(f"seq{si}:obs{t}:")}
            assert np.array_equal(g.agent_feat()[lo:hi].cpu().numpy(), ref["x_a"].astype(np.float32))
            for et, kx, ko in (("seen", "x_gt", "seen_off"), ("near", "x_ubs", "near_off")):
                x, off = g.relation_segments(et)
                e0, e1 = int(off[lo]), int(off[hi])
                assert np.array_equal((off[lo:hi + 1] - off[lo]).cpu().numpy(), ref[ko]), (si, t, et)
                assert np.array_equal(x[e0:e1].cpu().numpy(), ref[kx].astype(np.float32)), (si, t, et)
            off, src = g.talk_csc()
            e0, e1 = int(off[lo]), int(off[hi])
            assert np.array_equal((off[lo:hi + 1] - off[lo]).cpu().numpy(), ref["talk_off"])
            assert np.array_equal((src[e0:e1] - lo).cpu().numpy(), ref["talk_src"])
        h_ref, T1 = z[f"seq{si}:h"], T + 1
        assert np.array_equal(rb.mem["h"][si].cpu().numpy(), h_ref.astype(np.float32)) and h_ref.shape[0] == T1
        assert np.array_equal(rb.mem["state"][si].cpu().numpy(), z[f"seq{si}:state"])
        assert np.array_equal(b["acts"][:, si * n:(si + 1) * n].cpu().numpy(), z[f"seq{si}:act"])
        assert np.array_equal(b["rews"][:, si].cpu().numpy(), z[f"seq{si}:rew"].reshape(T, n))
        assert np.array_equal(b["dones"][:, si].cpu().numpy(), z[f"seq{si}:done"].reshape(T, 1))
    # h0 / h1 = the stored hidden states of the first two steps of every sequence (learner.py:113)
    assert np.array_equal(b["h0"].cpu().numpy().reshape(4, n, H), np.stack([z[f"seq{s}:h"][0] for s in range(4)]).astype(np.float32))
    assert np.array_equal(b["h1"].cpu().numpy().reshape(4, n, H), np.stack([z[f"seq{s}:h"][1] for s in range(4)]).astype(np.float32))
    # the episode ended inside sequence 3 (push 10 of 14): the hidden state after it is zero, the observation a reset
    assert float(np.abs(z["seq3:h"][1]).max()) == 0.0 and float(np.abs(z["seq3:h"][0]).max()) > 0.0


def _qmix_learner(device, dtype):
    """uav_bs_ctrl_amd learner with mixer=True loaded with the closed-form weights of learner_update_qmix.npz."""
    import ast
    from uav_bs_ctrl_amd import HeteroBatch
    from uav_bs_ctrl_amd.learner import MultiAgentQLearner
    z = np.load(f"{GOLDEN}/learner_update_qmix.npz")
    cfg = ast.literal_eval(str(z["cfg"]))
    args = types.SimpleNamespace(device=device, hidden_size=32, c="tarmac", n_heads=4, n_layers=2, msg_size=8, key_size=4,
                                 n_rounds=1, dueling=False, mixer=True, embed_dim=cfg["embed_dim"], double_q=True,
                                 lr=cfg["lr"], gamma=cfg["gamma"], polyak=cfg["polyak"], max_seq_len=cfg["T"], seed=0)
    env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=cfg["n_actions"], n_agents=cfg["n_agents"],
                    episode_limit=10, state_shape=cfg["state_dim"])
    L = MultiAgentQLearner(env_info, args)

    def fill(mod, names, shapes):
        assert [k.removeprefix("inner.") for k, _ in mod.named_parameters()] == [str(k) for k in names]
        with th.no_grad():
            for i, (k, p) in enumerate(mod.named_parameters()):
                assert repr(tuple(p.shape)) == str(shapes[i])
                p.copy_(closed_form_tensor(p.shape, 1.0 + i * math.pi / 7, 0.1 if p.dim() == 1 else 0.25, th.float64))
    fill(L.policy_net, z["param_names"], z["param_shapes"])
    fill(L.mixer, z["mixer_param_names"], z["mixer_param_shapes"])
    L.target_net.load_state_dict(L.policy_net.state_dict())
    L.target_mixer.load_state_dict(L.mixer.state_dict())
    obs = []
    for t in range(cfg["T"] + 1):
        g = {k.split(":")[1]: th.as_tensor(z[k]) for k in z.files if k.startswith(f"t{t}:")}
        g = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in g.items()}
        obs.append(HeteroBatch.from_arrays(**g).to(device))
    f = lambda k: th.as_tensor(z[k]).to(dtype).to(device)  # noqa: E731
    batch = dict(obs=obs, h0=f("h0"), h1=f("h1"), acts=th.as_tensor(z["acts"]).long().to(device), rews=f("rews"),
                 dones=f("dones"), states=f("states"))
    return L, batch, cfg, z


def test_qmix_learner_fixture_is_consistent_with_the_oracle_on_cpu():
    """The mixer=True update fixture captured from the reference learner: the torch QMixer + the CPU oracle agent
    reproduce its loss (the -m gpu twin runs the HIP agent against the same fixture)."""
    import uav_bs_ctrl_amd.learner as LM
    from tests.test_dp_gloo import OracleAgent
    saved = LM.agent_REGISTRY
    LM.agent_REGISTRY = {"gnn": OracleAgent}
    try:
        L, batch, cfg, z = _qmix_learner("cpu", th.float32)
        loss, _, _ = L.loss(batch)
        assert_close(loss, th.as_tensor(z["loss"]).double(), 1e-5, "QMIX LossQ (oracle agent)")
    finally:
        LM.agent_REGISTRY = saved


def test_stage_obs_then_push_equals_one_push():
    """stage_obs (observation half, before the simulator is stepped in place) + push (act / rew / done / next_*) leave the
    same memory as one push of the whole transition."""
    rng = np.random.default_rng(1)
    E, n, M, H, T = 2, 3, 5, 4, 3
    mk = lambda: SequenceReplay(capacity=4, max_seq_len=T, n_agents=n, n_gts=M, hidden_size=H, n_envs=E, state_dim=3,  # noqa: E731
                                r_comm=0.5, device="cpu")
    a, b = mk(), mk()
    trs = [_transition(rng, E, n, M, H, t) for t in range(2 * T + 1)]
    obs_keys = ("gt", "ubs", "agent", "d_u2u", "h", "state")
    for t in range(2 * T):
        nxt = {"next_" + k: trs[t + 1][k] for k in obs_keys}
        a.push(dict(trs[t], **nxt))
        buf = {k: trs[t][k].clone() for k in obs_keys}              # the simulator's own buffers ...
        b.stage_obs(buf)
        for k in obs_keys:
            buf[k].fill_(-7.0)                                       # ... overwritten by the step before push() is called
        b.push(dict({k: trs[t][k] for k in ("act", "rew", "done")}, **nxt))
    assert len(a) == len(b) == 4 and a.head == b.head
    for k in a.mem:
        assert th.equal(a.mem[k], b.mem[k]), k
