"""World-size-2 data parallelism with the HIP agent on the GPU (BASELINE config 4 in miniature): two processes, each with
its own shard of env-graph sequences, both on cuda:0 (the box has one GPU and RCCL refuses two ranks on one device, so the
collective of THIS test is gloo over device tensors; the RCCL binding of the same call sites is covered at world size 1 by
test_gpu_parity.py and by the bench's --force-dist test).  What is checked: start-up broadcast, the mean of the two ranks'
flat gradient buffers equals the gradient of ONE process on the concatenated batch, replicas are bit-identical after the
fused clip + AdamW + polyak launch, and the step equals the single-process step on the concatenated batch.

Counterpart in the reference: utils/mpi_pytorch.py:19-35 (mpi_avg_grads / sync_params; unused by its launchers)."""
import os
import socket
import types

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.gpu_util import synth_graph
from tests.util import wait_worker
from uav_bs_ctrl_amd import HeteroBatch, batch as hb_batch

pytestmark = pytest.mark.gpu
N_AG, M_GT, T, H, B = 4, 12, 4, 64, 48


def _args():
    return types.SimpleNamespace(device="cuda", hidden_size=H, c="tarmac", n_heads=4, n_layers=1, msg_size=16, key_size=8,
                                 n_rounds=1, dueling=False, mixer=False, double_q=True, lr=1e-3, gamma=0.99,
                                 polyak=0.9, max_seq_len=None, seed=0)


def _make_batch(seed, n_graphs):
    obs = [HeteroBatch.from_arrays(**synth_graph(n_graphs, N_AG, M_GT, "ragged", seed=seed * 100 + t, talk="sparse")).to("cuda")
           for t in range(T + 1)]
    gen = th.Generator().manual_seed(seed)
    N = n_graphs * N_AG
    cu = lambda x: x.cuda()  # noqa: E731
    return dict(obs=obs, h0=cu(0.1 * th.randn(N, H, generator=gen)), h1=cu(0.1 * th.randn(N, H, generator=gen)),
                acts=cu(th.randint(5, (T, N, 1), generator=gen)), rews=cu(th.rand(T, n_graphs, N_AG, generator=gen)),
                dones=cu((th.rand(T, n_graphs, 1, generator=gen) < 0.2).float()))


def _concat(b0, b1):
    obs = [hb_batch([a, b]) for a, b in zip(b0["obs"], b1["obs"])]
    cat = lambda k, d: th.cat([b0[k], b1[k]], d)  # noqa: E731
    return dict(obs=obs, h0=cat("h0", 0), h1=cat("h1", 0), acts=cat("acts", 1), rews=cat("rews", 1), dones=cat("dones", 1))


def _learner(seed):
    from uav_bs_ctrl_amd.learner import MultiAgentQLearner
    th.manual_seed(seed)
    env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=5, n_agents=N_AG, episode_limit=T)
    return MultiAgentQLearner(env_info, _args())


def _flat(params):
    return th.cat([p.detach().reshape(-1) for p in params])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    th.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = _learner(123 + rank)                                    # different initial weights per rank on purpose
        assert next(L.policy_net.parameters()).is_cuda and L.grads.flat.is_cuda
        flat0 = _flat(L.policy_net.parameters())
        gathered = [th.zeros_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, flat0)
        same_init = all(th.equal(gathered[0], g) for g in gathered)
        mine = _make_batch(10 + rank, B)
        L.grads.zero_()
        loss, _, _ = L.loss(mine)
        loss.backward()
        L.grads.all_reduce_mean_()
        grad = L.grads.flat.clone()
        out = L.update(mine)                                        # accumulate -> collective -> fused apply
        flat1 = _flat(L.policy_net.parameters())
        gathered = [th.zeros_like(flat1) for _ in range(world)]
        dist.all_gather(gathered, flat1)
        same_after = all(th.equal(gathered[0], g) for g in gathered)
        th.cuda.synchronize()
        if rank == 0:
            q.put(dict(same_init=same_init, same_after=same_after, grad=grad.cpu().numpy(), init=flat0.cpu().numpy(),
                       after=flat1.cpu().numpy(), target=_flat(L.target_net.parameters()).cpu().numpy(),
                       loss=float(out["LossQ"])))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_dp2_hip_agent_gradient_and_step_equal_single_process_on_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = wait_worker(procs[0], q, timeout=480)
    finally:
        for p in procs:
            p.join(60)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = {k: (th.as_tensor(v).cuda() if hasattr(v, "shape") else v) for k, v in res.items()}
    assert res["same_init"], "broadcast_parameters did not synchronise the replicas"
    assert res["same_after"], "replicas diverged after one update"

    # single process, concatenated batch, parameters = rank 0's initial parameters
    L = _learner(7)
    o = 0
    with th.no_grad():
        for p in L.policy_net.parameters():
            p.copy_(res["init"][o:o + p.numel()].view_as(p))
            o += p.numel()
        L.target_net.load_state_dict(L.policy_net.state_dict())
    both = _concat(_make_batch(10, B), _make_batch(11, B))
    L.grads.zero_()
    loss, _, _ = L.loss(both)
    loss.backward()
    ref = L.grads.flat.clone()
    ref_dense = th.cat([p.grad.reshape(-1) for p in L.policy_net.parameters()])   # parameters() order, as `after` below
    scale = float(ref.abs().max())
    err = float((res["grad"] - ref).abs().max())
    assert err <= 1e-5 * scale + 1e-9, f"DP-2 mean gradient != big-batch gradient: {err:.3e} at scale {scale:.3e}"
    out = L.update(both)
    assert abs(float(out["LossQ"]) - res["loss"]) <= 1.0   # ranks report their OWN shard's loss; same order of magnitude
    after = _flat(L.policy_net.parameters())
    # AdamW's first step moves every weight by ~lr whatever the gradient's size: entries whose gradient is rounding noise
    # may flip direction, so compare where the reference gradient is resolved and bound the rest by 2 lr
    step_err = (res["after"] - after).abs()
    assert float(step_err.max()) <= 2.1e-3
    resolved = ref_dense.abs() > 1e-3 * scale
    assert float(step_err[resolved].max()) <= 2e-6, "DP-2 step != single-process step"
    targ = _flat(L.target_net.parameters())
    assert float((res["target"] - targ).abs()[resolved].max()) <= 2e-6


@pytest.mark.timeout(1500)
def test_bench_at_world_size_two_on_one_device():
    """bench.py launched as the driver launches it for N = 2 (torch.distributed.run, two ranks, --gpus 2), both ranks mapped to
    cuda:0 over gloo by the file's test hooks: per-rank shards (seed 1234 + rank), barrier-bracketed timing with the max over
    ranks, the in-file assertion that the parameter replicas stay bit-identical, one JSON line from rank 0 with the whole-job
    aggregate.  The two ranks see different batches, so the step must differ from the one-rank step on rank 0's batch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--steps", "2", "--warmup", "1", "--B", "64", "--T", "4", "--dist", "env"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", UAVGNN_BENCH_BACKEND="gloo", UAVGNN_BENCH_ONE_DEVICE="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", *small],
                         cwd=root, env=env, capture_output=True, text=True, timeout=800)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]                      # ONE line, from rank 0
    r2 = json.loads(lines[0])
    assert r2["n_gpus"] == 2 and r2["config"]["global_batch"] == 128 and r2["config"]["parallelism"] == "dp2"
    assert r2["scaling"] == "weak" and r2["steps"] == 2 and r2["warmup"] == 1
    assert abs(r2["value"] - 2 * 64 * 4 * 2 / (r2["ms_per_step"] * 2e-3)) <= 1e-6 * r2["value"]
    assert "roofline" in r2 and "cpu_baseline" not in r2 and "end_to_end" not in r2
    assert r2["replicas_identical"] is True and r2["self_launched"] is False and r2["rccl_ranks"] == 2
    # the PLAIN form: `python3 bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment launches its own two ranks and gives
    # the same job (same shards, same arithmetic: identical parameter checksum and loss)
    plain_env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    plain = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *small], cwd=root, env=plain_env, capture_output=True,
                           text=True, timeout=800)
    assert plain.returncode == 0, plain.stderr[-3000:]
    lines = [ln for ln in plain.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, plain.stdout[-2000:]
    rp = json.loads(lines[0])
    assert rp["n_gpus"] == 2 and rp["self_launched"] is True and rp["replicas_identical"] is True
    # (two processes share the one GPU of the box in both runs: the same job to fp32 rounding - the vendor GEMMs' split-K order
    # is not pinned under that contention; the one-process determinism tests are test_gpu_parity.py's)
    assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(rp["params_checksum"], r2["params_checksum"]))
    assert abs(rp["loss"] - r2["loss"]) <= 1e-5 * abs(r2["loss"])
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", *small, "--no-cpu-baseline", "--no-end-to-end",
                          "--no-fp32-leg", "--no-rho-leg"], cwd=root, env=dict(os.environ), capture_output=True, text=True,
                         timeout=800)
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert r1["n_gpus"] == 1 and r1["params_checksum"] != r2["params_checksum"]
