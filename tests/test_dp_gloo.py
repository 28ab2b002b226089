"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel path of uav_bs_ctrl_amd/learner.py:
the flat-buffer gradient all-reduce reproduces the single-process gradient on the concatenated batch, replicas stay
bit-identical after an update, start-up broadcast works.  The HIP agent cannot run on CPU (no fallback), so the
processes use an nn.Module with the SAME parameter layout whose forward is the CPU oracle (test infrastructure)."""
import os
import socket
import types

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import restatement as R
from tests.gpu_util import synth_graph
from tests.util import wait_worker
from uav_bs_ctrl_amd import GnnAgent, HeteroBatch, batch as hb_batch

CFG = dict(enc="gnn", c="tarmac", n_heads=4, key_size=4, msg_size=8, n_rounds=1, dueling=False)
N_AG, M_GT, T, H = 3, 6, 3, 32


def _args():
    return types.SimpleNamespace(device="cpu", hidden_size=H, c="tarmac", n_heads=4, n_layers=1, msg_size=8, key_size=4,
                                 n_rounds=1, dueling=False, mixer=False, double_q=True, lr=1e-3, gamma=0.99,
                                 polyak=0.9, max_seq_len=None, seed=0)


class OracleAgent(nn.Module):
    """Parameter layout of GnnAgent, arithmetic of oracle/restatement.py."""

    def __init__(self, obs_shape, n_actions, args):
        super().__init__()
        self.inner = GnnAgent(obs_shape, n_actions, args)

    def init_hidden(self):
        return self.inner.init_hidden()

    def forward(self, g: HeteroBatch, h):
        arrays = dict(x_a=g.agent_feat())
        arrays["x_gt"], arrays["seen_off"] = g.relation_segments("seen")
        arrays["x_ubs"], arrays["near_off"] = g.relation_segments("near")
        arrays["talk_off"], arrays["talk_src"] = g.talk_csc()
        p = {k: v for k, v in self.inner.named_parameters()}
        return R.gnn_agent_forward(arrays, h, p, CFG)


def _make_batch(seed, B):
    obs = [HeteroBatch.from_arrays(**synth_graph(B, N_AG, M_GT, "ragged", seed=seed * 100 + t, talk="sparse"))
           for t in range(T + 1)]
    gen = th.Generator().manual_seed(seed)
    N = B * N_AG
    return dict(obs=obs, h0=0.1 * th.randn(N, H, generator=gen), h1=0.1 * th.randn(N, H, generator=gen),
                acts=th.randint(5, (T, N, 1), generator=gen), rews=th.rand(T, B, N_AG, generator=gen),
                dones=(th.rand(T, B, 1, generator=gen) < 0.2).float())


def _concat(b0, b1):
    obs = [hb_batch([a, b]) for a, b in zip(b0["obs"], b1["obs"])]
    cat = lambda k, d: th.cat([b0[k], b1[k]], d)  # noqa: E731
    return dict(obs=obs, h0=cat("h0", 0), h1=cat("h1", 0), acts=cat("acts", 1), rews=cat("rews", 1),
                dones=cat("dones", 1))


def _learner():
    import uav_bs_ctrl_amd.learner as LM
    LM.agent_REGISTRY = {"gnn": OracleAgent}
    th.manual_seed(123 + (dist.get_rank() if dist.is_initialized() else 0))   # different init per rank on purpose
    env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=5, n_agents=N_AG, episode_limit=T)
    return LM.MultiAgentQLearner(env_info, _args())


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    th.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = _learner()
        # start-up broadcast made the replicas identical although the seeds differ
        flat0 = th.cat([p.detach().reshape(-1) for p in L.policy_net.parameters()])
        gathered = [th.zeros_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, flat0)
        same_init = all(th.equal(gathered[0], g) for g in gathered)
        mine = _make_batch(10 + rank, B=2)
        L.grads.zero_()
        loss, _, _ = L.loss(mine)
        loss.backward()
        L.grads.all_reduce_mean_()
        grad = L.grads.flat.clone()
        out = L.update(mine)                                  # full step incl. clip / AdamW / polyak
        flat1 = th.cat([p.detach().reshape(-1) for p in L.policy_net.parameters()])
        tflat = th.cat([p.detach().reshape(-1) for p in L.target_net.parameters()])
        gathered = [th.zeros_like(flat1) for _ in range(world)]
        dist.all_gather(gathered, flat1)
        same_after = all(th.equal(gathered[0], g) for g in gathered)
        if rank == 0:
            # numpy (pickled by value): torch tensors travel by fd-sharing, which races with this process exiting
            q.put(dict(same_init=same_init, same_after=same_after, grad=grad.numpy(), init=flat0.numpy(),
                       after=flat1.numpy(), target=tflat.numpy(), loss=float(out["LossQ"])))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_dp2_gradient_equals_single_process_on_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = wait_worker(procs[0], q, timeout=240)
    res = {k: (th.as_tensor(v) if hasattr(v, "shape") else v) for k, v in res.items()}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["same_init"], "broadcast_parameters did not synchronise the replicas"
    assert res["same_after"], "replicas diverged after one update"

    # single process, concatenated batch, parameters = rank 0's initial parameters
    L = _learner()
    o = 0
    with th.no_grad():
        for p in L.policy_net.parameters():
            p.copy_(res["init"][o:o + p.numel()].view_as(p))
            o += p.numel()
        L.target_net.load_state_dict(L.policy_net.state_dict())
    both = _concat(_make_batch(10, 2), _make_batch(11, 2))
    L.grads.zero_()
    loss, _, _ = L.loss(both)
    loss.backward()
    ref = L.grads.flat.clone()
    scale = float(ref.abs().max())
    assert float((res["grad"] - ref).abs().max()) <= 1e-5 * scale + 1e-9, "DP-2 mean gradient != big-batch gradient"
    L.update(both)
    after = th.cat([p.detach().reshape(-1) for p in L.policy_net.parameters()])
    assert float((res["after"] - after).abs().max()) <= 1e-6, "DP-2 step != single-process step"
    targ = th.cat([p.detach().reshape(-1) for p in L.target_net.parameters()])
    assert float((res["target"] - targ).abs().max()) <= 1e-6


def test_flat_grad_buffer_views_and_clip():
    from uav_bs_ctrl_amd.learner import FlatGradBuffer
    net = nn.Sequential(nn.Linear(3, 4), nn.Linear(4, 2))
    fb = FlatGradBuffer(list(net.parameters()))
    net(th.ones(5, 3)).sum().backward()
    assert fb.flat.abs().sum() > 0
    ptrs = [p.grad.data_ptr() for p in net.parameters()]
    assert ptrs[0] == fb.flat.data_ptr() and ptrs == sorted(ptrs)
    net.zero_grad(set_to_none=True)          # something detached the views ...
    fb.zero_()                               # ... zero_() re-attaches them
    assert all(p.grad is not None and p.grad.data_ptr() == q for p, q in zip(net.parameters(), ptrs))
    assert float(fb.flat.abs().sum()) == 0.0


def test_replay_ratio_chunks_accumulate_to_the_concatenated_batch_step():
    """learner.update([chunk_0, .., chunk_{rho-1}]) (gradient accumulation, the reference's replay ratio as chunks of B
    sequences) takes the same step as ONE update on the concatenated batch: same loss, same parameters afterwards."""
    th.manual_seed(5)
    L1, L2 = _learner(), _learner()
    L2.policy_net.load_state_dict(L1.policy_net.state_dict())
    L2.target_net.load_state_dict(L1.target_net.state_dict())
    b0, b1 = _make_batch(20, 2), _make_batch(21, 2)
    out1 = L1.update([b0, b1])
    out2 = L2.update(_concat(b0, b1))
    assert abs(float(out1["LossQ"]) - float(out2["LossQ"])) <= 1e-6 * max(1.0, abs(float(out2["LossQ"])))
    for (k, p1), (_, p2) in zip(L1.policy_net.named_parameters(), L2.policy_net.named_parameters()):
        assert float((p1 - p2).abs().max()) <= 1e-6, k


def test_accumulate_rejects_empty_and_unequal_chunk_lists():
    """the mean of the chunk means equals the mean over all sequences only for equally sized chunks, and an empty
    list has no loss: both are refused instead of stepping on a silently different objective."""
    import pytest
    L1 = _learner()
    with pytest.raises(ValueError):
        L1.accumulate([])
    with pytest.raises(ValueError):
        L1.accumulate([_make_batch(1, 2), _make_batch(2, 3)])


def test_discrete_comm_noise_key_survives_a_checkpoint(tmp_path):
    """DiscreteComm's {seed, step} pair is not in the state_dict (its names are the reference's contract); the
    learner's checkpoint carries it as an extra key and load_checkpoint puts it back."""
    L1, L2 = _learner(), _learner()
    fake = type("M", (nn.Module,), {})()
    fake.rng_state = th.tensor([1234567, 89], dtype=th.int64)
    L1.policy_net.add_module("f_comm_probe", fake)
    probe2 = type("M", (nn.Module,), {})()
    probe2.rng_state = None
    L2.policy_net.add_module("f_comm_probe", probe2)
    path = str(tmp_path / "ck.pt")
    L1.save_checkpoint(path, dict(epoch=1, t=2))
    assert th.load(path)["comm_rng_state"]["f_comm_probe"].tolist() == [1234567, 89]
    L2.load_checkpoint(path)
    assert L2.policy_net.f_comm_probe.rng_state.tolist() == [1234567, 89]
