#!/usr/bin/env python
"""BUILD-CONTAINER-ONLY proof of INTEGRATION.md's drop-in claim (needs /root/reference; never runs on the GPU box).

The reference's OWN, UNCHANGED modules - envs/mubs_cov (simulator), algos/madrqn/utils/env_wrappers.py
(GraphObservation.build_obs_graph / local_observation, MultiUbsCoverageWrapper.observation / build_comm_graph),
algos/common.py (cat), algos/madrqn/buffer.py and algos/madrqn/learner.py (act / cache / update, incl. its
``p_targ.data`` polyak loop) - are imported with

    sys.modules['dgl'] = uav_bs_ctrl_amd.graph
    REGISTRY['gnn']    = an agent with uav_bs_ctrl_amd.GnnAgent's parameters

and driven through the exact recipe that produced tests/golden/learner_update_tarmac.npz (there the same reference code
ran over the DGL stand-in and its own GnnAgent).  Every graph the reference wrapper + cat now build as a HeteroBatch
must equal the fixture's arrays BIT-EXACTLY, and loss / Q-values / clipped gradients / post-step parameters / polyak
target must equal the fixture to 1e-9 (float64).  There is no GPU here and the HIP agent has no CPU path, so the agent's
arithmetic is supplied by the CPU oracle behind GnnAgent's parameter layout (tests/oracle_agent.py); the HIP arithmetic
is pinned to the same fixture by tests/test_gpu_parity.py::test_learner_update_reproduces_reference_update.
"""
import os
import random
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "oracle", "gym_standin"), REF, ROOT]

import uav_bs_ctrl_amd.graph as G  # noqa: E402

sys.modules["dgl"] = G                                             # `import dgl` in the reference now finds HeteroBatch
from tests.oracle_agent import OracleBackedAgent  # noqa: E402

shim = types.ModuleType("algos.madrqn.agents.gnn_agents")          # what the edited agents/__init__.py would import
shim.GnnAgent = OracleBackedAgent
sys.modules["algos.madrqn.agents.gnn_agents"] = shim

from algos.common import cat as ref_cat  # noqa: E402
from algos.madrqn.learner import MultiAgentQLearner  # noqa: E402
from algos.madrqn.utils.env_wrappers import MultiUbsCoverageWrapper  # noqa: E402
from envs.mubs_cov.mubs_cov import MultiUbsCoverageEnv  # noqa: E402
from oracle.closed_form import fill_closed_form  # noqa: E402

th.set_default_dtype(th.float64)
KEYS = ("x_a", "x_gt", "seen_off", "x_ubs", "near_off", "talk_off", "talk_src", "talk_eid")


def arrays_of(g):
    out = dict(x_a=g.agent_feat())
    out["x_gt"], out["seen_off"] = g.relation_segments("seen")
    out["x_ubs"], out["near_off"] = g.relation_segments("near")
    out["talk_off"], out["talk_src"] = g.talk_csc()
    out["talk_eid"] = g.talk_eid()
    return {k: v.numpy() for k, v in out.items()}


def to_double(g):
    g.ndata["feat"] = {nt: f.double() for nt, f in g.ndata["feat"].items()}
    return g


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "learner_update_tarmac.npz"))
    T, B = 5, 4
    args = types.SimpleNamespace(device="cpu", o="gnn", c="tarmac", hidden_size=32, n_heads=4, n_layers=2, msg_size=8,
                                 key_size=4, n_rounds=1, dueling=False, mixer=False, max_seq_len=T, gamma=0.99,
                                 polyak=0.995, batch_size=B, replay_size=100, lr=5e-4, anneal_lr=False, double_q=True,
                                 share_reward=False, norm_r=False)
    np.random.seed(11), random.seed(11), th.manual_seed(11)
    env = MultiUbsCoverageWrapper(MultiUbsCoverageEnv("debug", record=False), args)
    learner = MultiAgentQLearner(env.get_env_info(), args)
    assert isinstance(learner.policy_net, OracleBackedAgent)
    fill_closed_form(learner.policy_net.inner)
    learner.target_net.load_state_dict(learner.policy_net.state_dict())
    random.seed(12), th.manual_seed(12)
    (o, s), h = env.reset(), learner.init_hidden()
    assert isinstance(o, G.HeteroBatch) and o.graph_off.tolist() == [0, env.n_agents]
    o = to_double(o)
    while len(learner.buffer) < B:
        a, h2 = learner.act(o, h, 0.5)
        o2, s2, r, d, info = env.step(a)
        o2 = to_double(o2)
        learner.cache(o, h, s, a, r, o2, h2, s2, d, info.get("BadMask"))
        o, s, h = o2, s2, h2
        if d:
            (o, s), h = env.reset(), learner.init_hidden()
            o = to_double(o)
    samples = list(learner.buffer.memory)[:B]
    learner.buffer.sample = lambda n: samples
    n_cmp = 0
    for tt in range(T + 1):
        g = ref_cat([samples[i]["obs"][tt] for i in range(B)])            # algos.common.cat -> HeteroBatch batch
        got = arrays_of(g)
        for k in KEYS:
            ref = z[f"t{tt}:{k}"]
            assert got[k].shape == ref.shape and np.array_equal(got[k], ref.astype(got[k].dtype)), (tt, k)
            n_cmp += 1
        assert g.graph_off.tolist() == list(range(0, B * env.n_agents + 1, env.n_agents))
    res = learner.update()                                                # reference update(), unchanged
    tol = 1e-9
    assert abs(res["LossQ"] - float(z["loss"])) < tol, (res["LossQ"], float(z["loss"]))
    assert np.abs(res["QVals"] - z["qvals"]).max() < tol
    worst = 0.0
    for k, p in learner.policy_net.inner.named_parameters():
        worst = max(worst, float(np.abs(p.grad.numpy() - z["grad_clipped:" + k]).max()),
                    float(np.abs(p.detach().numpy() - z["policy_after:" + k]).max()))
    for k, p in learner.target_net.inner.named_parameters():
        worst = max(worst, float(np.abs(p.detach().numpy() - z["target_after:" + k]).max()))
    assert worst < tol, worst
    print(f"DROPIN OK: {n_cmp} graph arrays bit-exact; loss {res['LossQ']:.9f} (fixture {float(z['loss']):.9f}); "
          f"max |grad/param/target - fixture| = {worst:.2e}")


if __name__ == "__main__":
    main()
