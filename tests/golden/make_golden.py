#!/usr/bin/env python
"""Generates tests/golden/*.npz by running the REFERENCE's own modules (imported unchanged from /root/reference)
over oracle/dgl_standin + oracle/gym_standin.  Runs only in the build container (the reference cannot travel).

    python tests/golden/make_golden.py            # rewrites every fixture

Each fixture holds: segment-layout graph arrays, h, closed-form weights (state_dict of the reference module),
outputs (q, h') and gradients of  L = sum(q*wq) + sum(h'*wh)  w.r.t. every parameter and h, all float64
(the tests down-cast as needed).  Fixtures are data only - no reference source text is stored.

What is pinned: the reference's wiring (gnn_agents.py, dueling.py, env_wrappers.py graph layout, common.cat) on
top of a restated DGL ("parity unpinned" at the DGL boundary - see oracle/restatement.py header).
"""
import os
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "oracle", "dgl_standin"), os.path.join(ROOT, "oracle", "gym_standin"), REF, ROOT]

import dgl  # noqa: E402  (the stand-in)
from algos.common import cat as ref_cat  # noqa: E402
from algos.madrqn.agents import REGISTRY as MADRQN_REGISTRY  # noqa: E402
from algos.madrqn.utils.env_wrappers import GraphObservation, MultiUbsCoverageWrapper  # noqa: E402
from algos.drqn.agents.gnn_agents import GnnAgent as DrqnGnnAgent  # noqa: E402
from algos.drqn.utils.env_wrappers import GraphObservation as DrqnGraphObservation  # noqa: E402

from oracle.closed_form import closed_form_tensor, fill_closed_form  # noqa: E402

th.set_default_dtype(th.float64)


# ---------------------------------------------------------------------------------------------------------------------
def graph_to_arrays(g):
    """Stand-in DGLGraph -> segment-layout arrays (asserting the layout invariants of SURVEY A.0)."""
    out = {}
    n = g.num_nodes("agent")
    feats = g.ndata["feat"]
    out["x_a"] = feats["agent"].numpy()
    for et, st, key, off in (("seen", "gt", "x_gt", "seen_off"), ("near", "ubs", "x_ubs", "near_off")):
        if not any(c[1] == et for c in g.canonical_etypes):
            continue
        src, dst = g.edges(et)
        assert th.equal(src, th.arange(src.numel())), "src id != edge id"
        assert bool((dst[1:] >= dst[:-1]).all()), "edges not grouped by destination"
        assert g.num_nodes(st) == src.numel()
        deg = th.bincount(dst, minlength=n)
        out[off] = np.concatenate([[0], np.cumsum(deg.numpy())]).astype(np.int32)
        out[key] = feats[st].numpy() if st in feats else np.zeros((0, 1))
    if any(c[1] == "talk" for c in g.canonical_etypes):
        src, dst = g.edges("talk")
        order = th.sort(dst, stable=True)[1]
        deg = th.bincount(dst, minlength=n)
        out["talk_off"] = np.concatenate([[0], np.cumsum(deg.numpy())]).astype(np.int32)
        out["talk_src"] = src[order].numpy().astype(np.int32)
        out["talk_eid"] = order.numpy().astype(np.int32)     # CSC position -> reference edge id
    return out


def synth_obs(rng, n, M, deg_seen, deg_near):
    """Per-agent obs dicts in the env's format (mubs_cov.py:215-242): column 0 is the visibility flag."""
    obs = []
    for i in range(n):
        gt = np.zeros((M, 5), dtype=np.float64)
        vis = rng.choice(M, size=deg_seen[i], replace=False)
        gt[vis, 0] = 1
        gt[:, 1:3] = rng.uniform(-1, 1, (M, 2))
        gt[:, 3:5] = rng.uniform(0, 1, (M, 2))
        ubs = np.zeros((n - 1, 3), dtype=np.float64)
        visu = rng.choice(n - 1, size=deg_near[i], replace=False) if n > 1 else []
        ubs[visu, 0] = 1
        ubs[:, 1:3] = rng.uniform(-1, 1, (n - 1, 2))
        obs.append(dict(agent=rng.uniform(0, 1, 2), ubs=ubs, gt=gt))
    return obs


def ref_env_graph(obs, d_u2u, r_comm, with_comm=True):
    """One env-step graph, built by the reference's own wrapper code (env_wrappers.py:65-89,:122-154)."""
    local = GraphObservation.local_observation(GraphObservation.__new__(GraphObservation), obs)
    if not with_comm:
        return local
    fake = types.SimpleNamespace(n_agents=len(obs), d_u2u=d_u2u, r_comm=r_comm)
    comm = MultiUbsCoverageWrapper.build_comm_graph(fake)
    return dgl.merge([local, comm])


def synth_batch(seed, with_comm=True):
    """3 envs x 4 agents, M = 80; seen degrees cover {0,1,2,17,64,80}; talk: complete / sparse / self-loops only."""
    rng = np.random.default_rng(seed)
    n, M = 4, 80
    degs = [[0, 1, 2, 17], [64, 80, 0, 5], [3, 0, 80, 33]]
    degn = [[3, 0, 1, 2], [3, 3, 0, 1], [0, 2, 3, 1]]
    graphs = []
    for b in range(3):
        obs = synth_obs(rng, n, M, degs[b], degn[b])
        if b == 0:
            d = np.zeros((n, n))                      # complete incl. self loops
        elif b == 1:
            d = np.where(rng.uniform(size=(n, n)) < 0.4, 0.0, 10.0)
            np.fill_diagonal(d, 0.0)
        else:
            d = np.full((n, n), 10.0)
            np.fill_diagonal(d, 0.0)                  # self loops only
        graphs.append(ref_env_graph(obs, d, 1.0, with_comm))
    return ref_cat(graphs)


def make_args(c, H=32, dueling=False, n_rounds=1, n_layers=2):
    return types.SimpleNamespace(hidden_size=H, c=c, n_heads=4, n_layers=n_layers, msg_size=8, key_size=4,
                                 n_rounds=n_rounds, dueling=dueling)


def run_and_save(name, net, g, h, arrays, cfg, gumbel_seed=None):
    fill_closed_form(net)
    h = h.clone().requires_grad_(True)
    extra = {}
    if gumbel_seed is not None:
        # F.gumbel_softmax draws -log(Exp(1)) with one exponential_() call on a tensor shaped like the logits
        # [E, msg, 2] in reference edge-id order (gnn_agents.py:172); reproduce the same draw, store it in CSC order.
        E = g.number_of_edges("talk")
        th.manual_seed(gumbel_seed)
        gum = -th.empty(E, cfg["msg_size"], 2).exponential_().log()
        extra["gumbel"] = gum[th.as_tensor(arrays["talk_eid"]).long()].numpy()
        th.manual_seed(gumbel_seed)
    q, h2 = net(g, h)
    wq = closed_form_tensor(tuple(q.shape), 101.0)
    wh = closed_form_tensor(tuple(h2.shape), 202.0)
    loss = (q * wq).sum() + (h2 * wh).sum()
    params = dict(net.named_parameters())
    grads = th.autograd.grad(loss, list(params.values()) + [h], allow_unused=True)
    out = dict(arrays)
    out.update(extra)
    out["h"] = h.detach().numpy()
    out["q"] = q.detach().numpy()
    out["h_out"] = h2.detach().numpy()
    out["wq"], out["wh"] = wq.numpy(), wh.numpy()
    for (k, p), gr in zip(list(params.items()) + [("__h__", h)], grads):
        out["grad:" + k] = (gr if gr is not None else th.zeros_like(p)).numpy()
    # parameters are NOT stored: tests rebuild them with oracle.closed_form (same index order)
    out["param_names"] = np.array(list(params.keys()))
    out["param_shapes"] = np.array([repr(tuple(p.shape)) for p in params.values()])
    out["cfg"] = np.array(repr(cfg))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N_a={q.shape[0]} q={tuple(q.shape)} |q|max={q.abs().max():.4f} -> {os.path.getsize(path)} B")


def main():
    obs_shape = dict(agent=2, ubs=2, gt=4)
    n_actions = 9
    # A. every comm variant on the ragged synthetic batch
    variants = [("none", None, {}), ("tarmac", "tarmac", {}), ("tarmac_r2", "tarmac", dict(n_rounds=2)),
                ("tarmac_duel", "tarmac", dict(dueling=True)), ("disc", "disc", {}), ("base", "base", {}),
                ("commnet", "commnet", dict(n_rounds=2)), ("econv", "econv", {})]
    for name, c, kw in variants:
        args = make_args(c, **kw)
        g = synth_batch(7, with_comm=True)
        arrays = graph_to_arrays(g)
        net = MADRQN_REGISTRY["gnn"](obs_shape, n_actions, args)
        h = closed_form_tensor((g.num_nodes("agent"), args.hidden_size), 303.0) * 5
        cfg = dict(enc="gnn", c=c, n_heads=4, key_size=4, msg_size=8, n_rounds=args.n_rounds, n_layers=args.n_layers,
                   dueling=args.dueling, hidden_size=32, n_actions=n_actions)
        run_and_save("agent_" + name, net, g, h, arrays, cfg, gumbel_seed=11 if c == "disc" else None)

    # B. dense (MLP) observation encoder + TarMAC (exp2 arm: o='mlp' with comm; gnn_agents.py:22-23,:62-77)
    args = make_args("tarmac", n_layers=2)
    g = synth_batch(7, with_comm=True)
    arrays = graph_to_arrays(g)
    n = g.num_nodes("agent")
    x_flat = closed_form_tensor((n, 23), 404.0) * 5
    comm_only = g["talk"]                      # agent->agent slice shares the agent frame
    g.nodes["agent"].data["feat"] = x_flat     # env_wrappers.py:134
    arrays = {k: v for k, v in arrays.items() if k.startswith("talk")}
    arrays["x_flat"] = x_flat.numpy()
    net = MADRQN_REGISTRY["gnn"](23, n_actions, args)
    h = closed_form_tensor((n, 32), 303.0) * 5
    cfg = dict(enc="mlp", c="tarmac", n_heads=4, key_size=4, msg_size=8, n_rounds=1, n_layers=2, dueling=False,
               hidden_size=32, n_actions=n_actions)
    del comm_only
    run_and_save("agent_mlp_tarmac", net, g, h, arrays, cfg)

    # C. DRQN twin (drqn/agents/gnn_agents.py:9-30): single relation, every GT connected, 4 GT features kept
    rng = np.random.default_rng(5)
    gs = []
    for _ in range(5):
        obs = dict(gt=rng.uniform(-1, 1, (20, 4)), agent=rng.uniform(0, 1, 2))
        gs.append(DrqnGraphObservation.observation(DrqnGraphObservation.__new__(DrqnGraphObservation), obs))
    g = dgl.batch(gs)
    src, dst = g.edges()
    arrays = dict(x_a=g.ndata["feat"]["agent"].numpy(), x_gt=g.ndata["feat"]["gt"].numpy(),
                  seen_off=np.concatenate([[0], np.cumsum(th.bincount(dst, minlength=5).numpy())]).astype(np.int32))
    assert th.equal(src, th.arange(src.numel()))
    args = types.SimpleNamespace(hidden_size=32, n_heads=4)
    net = DrqnGnnAgent(dict(agent=2, gt=4), 5, args)
    h = closed_form_tensor((5, 32), 303.0) * 5
    cfg = dict(enc="drqn", c=None, n_heads=4, hidden_size=32, n_actions=5)
    run_and_save("agent_drqn", net, g, h, arrays, cfg)

    # D. env-derived: the reference's deterministic Debug map (maps.py:38-50), 3 UBS x 4 GT, 4 steps batched
    from envs.mubs_cov.mubs_cov import MultiUbsCoverageEnv
    wargs = types.SimpleNamespace(o="gnn", c="tarmac", norm_r=False, share_reward=False)
    np.random.seed(3)
    env = MultiUbsCoverageWrapper(MultiUbsCoverageEnv("debug", record=False), wargs)
    o, _ = env.reset()
    frames = [o]
    for t in range(3):
        o, _, _, _, _ = env.step([(t + i) % env.n_actions for i in range(env.n_agents)])
        frames.append(o)
    g = ref_cat(frames)
    for fr in g._nframes.values():       # the env emits float32 features; fixtures are float64 throughout
        for k in list(fr):
            fr[k] = fr[k].double()
    arrays = graph_to_arrays(g)
    args = make_args("tarmac")
    net = MADRQN_REGISTRY["gnn"](env.get_obs_size(), env.n_actions, args)
    h = closed_form_tensor((g.num_nodes("agent"), 32), 303.0) * 5
    cfg = dict(enc="gnn", c="tarmac", n_heads=4, key_size=4, msg_size=8, n_rounds=1, n_layers=2, dueling=False,
               hidden_size=32, n_actions=env.n_actions)
    run_and_save("agent_debugmap_tarmac", net, g, h, arrays, cfg)

    make_learner_golden()
    make_mixer_golden()


def make_mixer_golden():
    """F. QMixer (algos/madrqn/agents/mixers.py:6-49): forward/backward on closed-form weights."""
    from algos.madrqn.agents.mixers import QMixer
    T, B, n, S = 3, 4, 5, 11
    mix = QMixer(S, n, types.SimpleNamespace(embed_dim=8))
    fill_closed_form(mix)
    qs = (closed_form_tensor((T * B, n), 0.3) * 10).view(T, B, n).clone().requires_grad_(True)
    st = (closed_form_tensor((T * B, S), 0.7) * 10).view(T, B, S)
    y = mix(qs, st)
    w = closed_form_tensor((T * B, 1), 1.9).view(T, B, 1)
    grads = th.autograd.grad((y * w).sum(), list(mix.parameters()) + [qs])
    out = dict(qs=qs.detach().numpy(), states=st.numpy(), y=y.detach().numpy(), w=w.numpy(),
               param_names=np.array([k for k, _ in mix.named_parameters()]),
               param_shapes=np.array([repr(tuple(p.shape)) for p in mix.parameters()]))
    for (k, _), g in zip(list(mix.named_parameters()) + [("__qs__", None)], grads):
        out["grad:" + k] = g.numpy()
    np.savez_compressed(os.path.join(HERE, "qmixer.npz"), **out)
    print("qmixer: y", tuple(y.shape))


def make_learner_golden():
    """E. Row L: the reference's own MultiAgentQLearner (algos/madrqn/learner.py) - act / cache / update - on the Debug
    map (3 UBS x 4 GT), H=32, T=5, B=4, anneal_lr=False (LambdaLR(verbose=True) is a TypeError on torch 2.10).  Stores
    the sampled batch (segment arrays per time step), initial hidden states, actions, rewards, dones and what update()
    produced: loss, Q-values, clipped gradients, post-step policy parameters and polyak-averaged target parameters."""
    import random

    from algos.madrqn.learner import MultiAgentQLearner
    from envs.mubs_cov.mubs_cov import MultiUbsCoverageEnv

    T, B = 5, 4
    args = types.SimpleNamespace(device="cpu", o="gnn", c="tarmac", hidden_size=32, n_heads=4, n_layers=2, msg_size=8,
                                 key_size=4, n_rounds=1, dueling=False, mixer=False, max_seq_len=T, gamma=0.99,
                                 polyak=0.995, batch_size=B, replay_size=100, lr=5e-4, anneal_lr=False, double_q=True,
                                 share_reward=False, norm_r=False)
    np.random.seed(11), random.seed(11), th.manual_seed(11)
    env = MultiUbsCoverageWrapper(MultiUbsCoverageEnv("debug", record=False), args)

    def to_double(g):
        for fr in g._nframes.values():
            for k in list(fr):
                fr[k] = fr[k].double()
        return g

    learner = MultiAgentQLearner(env.get_env_info(), args)
    fill_closed_form(learner.policy_net)
    learner.target_net.load_state_dict(learner.policy_net.state_dict())
    (o, s), h = env.reset(), learner.init_hidden()
    o = to_double(o)
    t = 0
    while len(learner.buffer) < B:
        a, h2 = learner.act(o, h, 0.5)
        o2, s2, r, d, info = env.step(a)
        o2 = to_double(o2)
        learner.cache(o, h, s, a, r, o2, h2, s2, d, info.get("BadMask"))
        o, s, h = o2, s2, h2
        t += 1
        if d:
            (o, s), h = env.reset(), learner.init_hidden()
            o = to_double(o)
    samples = list(learner.buffer.memory)[:B]
    learner.buffer.sample = lambda n: samples
    out = {}
    for tt in range(T + 1):
        g = ref_cat([samples[i]["obs"][tt] for i in range(B)])
        for k, v in graph_to_arrays(g).items():
            out[f"t{tt}:{k}"] = v
    out["h0"] = th.cat([samples[i]["h"][0] for i in range(B)]).numpy()
    out["h1"] = th.cat([samples[i]["h"][1] for i in range(B)]).numpy()
    out["acts"] = th.stack([th.cat([samples[i]["act"][tt] for i in range(B)]) for tt in range(T)]).numpy()
    out["rews"] = th.stack([th.cat([samples[i]["rew"][tt] for i in range(B)]) for tt in range(T)]).double().numpy()
    out["dones"] = th.stack([th.cat([samples[i]["done"][tt] for i in range(B)]) for tt in range(T)]).double().numpy()
    res = learner.update()
    out["loss"] = np.array(res["LossQ"])
    out["qvals"] = res["QVals"]
    names = [k for k, _ in learner.policy_net.named_parameters()]
    for k, p in learner.policy_net.named_parameters():
        out["grad_clipped:" + k] = p.grad.detach().numpy()
        out["policy_after:" + k] = p.detach().numpy()
    for k, p in learner.target_net.named_parameters():
        out["target_after:" + k] = p.detach().numpy()
    out["param_names"] = np.array(names)
    out["param_shapes"] = np.array([repr(tuple(p.shape)) for p in learner.policy_net.parameters()])
    out["cfg"] = np.array(repr(dict(enc="gnn", c="tarmac", n_heads=4, key_size=4, msg_size=8, n_rounds=1, n_layers=2,
                                    dueling=False, hidden_size=32, n_actions=env.n_actions, n_agents=env.n_agents, T=T,
                                    B=B, gamma=0.99, polyak=0.995, lr=5e-4, double_q=True)))
    path = os.path.join(HERE, "learner_update_tarmac.npz")
    np.savez_compressed(path, **out)
    print(f"learner_update_tarmac: loss={float(out['loss']):.6f} dones={out['dones'].sum():.0f} -> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
