"""Launch contract of bench.py at N > 1, on the CPU (no GPU here): both launch forms must start N ranks, rendezvous, and have rank 0
print ONE JSON line with ``n_gpus == N``.

  plain   ``python3 bench.py --gpus 2``                         - bench.py launches its ranks itself (self_launch)
  driver  ``python -m torch.distributed.run ... bench.py --gpus 2``

``UAVGNN_BENCH_DRY=1`` (a test hook of bench.py) replaces the device work by a trivial step and the RCCL group by gloo; everything
else - rank discovery, the re-exec, the barrier-bracketed max-over-ranks timing, the replica check, the one line - is the code the
GPU run executes.  The real two-rank HIP run of both forms is tests/test_dp_gpu.py (``-m gpu``).

Reference counterpart: utils/mpi_tools.py:6-36 (``mpi_fork``: re-exec under mpirun), utils/mpi_pytorch.py:19-35."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_line(proc):
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["UAVGNN_BENCH_DRY"] = "1"
    return env


@pytest.mark.timeout(600)
def test_plain_python_bench_gpus_2_launches_its_own_ranks():
    r = _one_line(subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=_env(),
                                 capture_output=True, text=True, timeout=500))
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["self_launched"] is True
    assert r["replicas_identical"] is True and r["steps"] == 3 and r["warmup"] == 1
    assert r["config"]["parallelism"] == "dp2" and r["config"]["global_batch"] == 2 * 4096


@pytest.mark.timeout(600)
def test_torch_distributed_run_form_still_works():
    r = _one_line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2",
                                  "--steps", "2", "--warmup", "0"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=500))
    assert r["n_gpus"] == 2 and r["self_launched"] is False and r["replicas_identical"] is True


@pytest.mark.timeout(300)
def test_rank_with_the_wrong_world_size_fails_with_instructions():
    env = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=250)
    assert p.returncode != 0 and "torch.distributed.run" in p.stderr and "launches its ranks itself" in p.stderr


def test_metric_string_follows_the_configuration():
    sys.path.insert(0, ROOT)
    import types

    import bench
    assert bench.metric_name(types.SimpleNamespace(n=8, M=80)) == "env-steps/sec (MADRQN, 8 UBS x 80 GT)"
    assert bench.metric_name(types.SimpleNamespace(n=4, M=40)) == "env-steps/sec (MADRQN, 4 UBS x 40 GT)"
    assert bench.metric_name(types.SimpleNamespace(n=16, M=200)) == "env-steps/sec (MADRQN, 16 UBS x 200 GT)"


def test_rccl_transport_summary_reads_an_nccl_debug_log(tmp_path):
    """`rccl_transport` of the bench line: transports counted from the `Channel ... via <transport>` lines of an NCCL_DEBUG=INFO log."""
    sys.path.insert(0, ROOT)
    import bench
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "box:101:201 [0] NCCL INFO NET/Plugin: Failed to find ncclNetPlugin_v8 symbol.",
        "box:101:201 [0] NCCL INFO comm 0x55 rank 0 nRanks 8 nNodes 1 localRanks 8 localRank 0 MNNVL 0",
        "box:101:201 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC comm 0x55 nRanks 08",
        "box:101:201 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC comm 0x55 nRanks 08",
        "box:101:201 [0] NCCL INFO Channel 00 : 0[c000] -> 7[e000] via P2P/direct pointer",
        "box:101:201 [0] NCCL INFO Connected all rings",
        "box:101:201 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1",
    ]))
    s = bench.rccl_transport_summary(str(log))
    assert s["channel_transports"] == {"P2P/IPC": 2, "P2P/direct pointer": 1}
    assert s["log_lines"] == 7 and any("Connected all rings" in ln for ln in s["lines"])
    assert bench.rccl_transport_summary(str(tmp_path / "missing.log")) is None
