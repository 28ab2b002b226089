#!/usr/bin/env python
"""bench.py - env-steps/s of the MADRQN hetero-GNN hot path on MI355X (BASELINE.json metric), one JSON line.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload, SURVEY 8d "C3"): 8 UBS x 80 GT env graphs, exp3 model (H=256, 4 heads, TarMAC msg 64 /
key 16, 9 actions), B = 4096 env graphs per GPU, degree distribution D-dense (every agent sees all 80 GTs, 7 neighbours,
complete talk graph incl. self loops), synthetic features, default (random) initial weights.

One "step" = one MADRQN cycle at replay ratio 1 over the B environments of a rank (SURVEY 8d iii):
    T = 50 rollout forwards (``learner.act``: no_grad policy forward + epsilon-greedy, B graphs each)  followed by
    ONE ``learner.update`` on B stored sequences of T transitions: 2T+1 = 101 forwards (policy with grad, target
    without), double-Q MSE loss, BPTT backward through T+1 steps, gradient all-reduce (N > 1), clip, AdamW, polyak.
Such a cycle advances B*T environment steps, so  value = N * B * T * K / elapsed  (whole-job env-steps/s, weak scaling:
B per GPU fixed).  The simulator itself is out of scope: graphs are synthetic and resident in HBM
before the timed region, exactly as the reference's sampled batch is when ``update`` starts its forward passes.

Extra objects: ``roofline`` for the dominant message-passing kernel (K1 forward, BOTH relations in one fused launch),
measured live with HIP events around every one of its launches inside the timed region - reported against the roof that binds at the
launch mix's arithmetic intensity (fp32 MFMA peak for the dense workload, HBM for env-realistic degrees), with both
fractions kept side by side (``hbm``, ``mfma_fp32``); ``cpu_baseline`` = the CPU oracle
(oracle/restatement.py, kind "port": the reference's DGL path cannot run here or on the GPU box) timed on the host
cores on a bounded sample of the same cycle.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import re
import sys
import time
import types

import torch as th
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3   # fp32 vector = fp32 MFMA peak
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; an MFMA-only loop sustains 2080-2140, 1840 on random operands)


def exp3_args(device, c="tarmac"):
    # run_exp3.py:30-54 + madrqn/config.py: H=256, 4 heads, msg 64, key 16, double_q, no dueling/mixer
    return types.SimpleNamespace(device=device, hidden_size=256, c=c, n_heads=4, n_layers=2, msg_size=64, key_size=16,
                                 n_rounds=1, dueling=False, mixer=False, double_q=True, lr=5e-4, gamma=0.99,
                                 polyak=0.999, max_seq_len=None, batch_size=None, seed=0)


def synth_batch_gpu(B, n, M, dist_name, device, gen):
    """Synthetic batched env graphs, generated directly in HBM (features U(-1,1)/U(0,1) as SURVEY 8d)."""
    from uav_bs_ctrl_amd import HeteroBatch
    N = B * n
    if dist_name == "dense":
        d_seen = th.full((N,), M, dtype=th.int64, device=device)
    else:  # D-env: 94 % of agents see nothing, the rest U{1..0.65 M}
        hi = max(1, int(0.65 * M))
        z = th.rand(N, device=device, generator=gen) < {"zero": 2.0, "nz": -1.0}.get(dist_name, 0.94)
        d_seen = th.where(z, th.zeros_like(z, dtype=th.int64), th.randint(1, hi + 1, (N,), device=device, generator=gen))
    seen_off = th.zeros(N + 1, dtype=th.int32, device=device)
    seen_off[1:] = th.cumsum(d_seen, 0).to(th.int32)
    near_off = th.arange(0, N * (n - 1) + 1, n - 1, dtype=th.int32, device=device)
    Es, En = int(seen_off[-1]), N * (n - 1)
    x_gt = th.rand(Es, 4, device=device, generator=gen) * 2 - 1
    x_gt[:, 2:] = th.rand(Es, 2, device=device, generator=gen)
    x_ubs = th.rand(En, 2, device=device, generator=gen) * 2 - 1
    x_a = th.rand(N, 2, device=device, generator=gen)
    talk_off = th.arange(0, N * n + 1, n, dtype=th.int32, device=device)
    base = (th.arange(N, device=device) // n * n).repeat_interleave(n)
    talk_src = (base + th.arange(n, device=device).repeat(N)).to(th.int32)
    return HeteroBatch.from_arrays(x_a=x_a, x_gt=x_gt, seen_off=seen_off, x_ubs=x_ubs, near_off=near_off,
                                   talk_off=talk_off, talk_src=talk_src,
                                   graph_off=th.arange(0, N + 1, n, dtype=th.int32, device=device), device=device,
                                   hints={"max_graph_agents": n, "max_deg:seen": M, "max_deg:near": n - 1})


def make_sequence(B, n, M, T, dist_name, device, seed, distinct):
    gen = th.Generator(device=device)
    gen.manual_seed(seed)
    graphs = [synth_batch_gpu(B, n, M, dist_name, device, gen) for _ in range(distinct)]
    obs = [graphs[t % distinct] for t in range(T + 1)]
    N = B * n
    # the sampled batch in time-major order, as the tensor-native replay hands it over (uav_bs_ctrl_amd/replay.py):
    # all T+1 observation graphs as ONE graph for the time-batched encoder (steps 1..T again for the target net)
    from uav_bs_ctrl_amd import batch as hb_batch
    obs_all, obs_all_next = hb_batch(obs), hb_batch(obs[1:])
    batch = dict(obs=obs, obs_all=obs_all, obs_all_next=obs_all_next,
                 h0=th.zeros(N, 256, device=device), h1=0.1 * th.randn(N, 256, device=device, generator=gen),
                 acts=th.randint(9, (T, N, 1), device=device, generator=gen),
                 rews=th.rand(T, B, n, device=device, generator=gen),
                 dones=th.zeros(T, B, 1, device=device))
    batch["dones"][-1] = 1.0
    return batch


def alg_k1_fwd(E_s, E_n, N, save_s, save_n):
    """ALGORITHMIC bytes / flops of one K1-forward launch over BOTH relations: per
    `seen` edge 16 B (x_gt row) and 3360 flop, per `near` edge 8 B and 2320 flop, per agent 8 B (x_a) + 8 B (two
    offsets) + 2048 B (the [2H] output row) and 7168 flop; + 16 B per edge of saved attention weights when the launch is
    part of a training forward."""
    b = 16 * E_s + 8 * E_n + N * (8 + 8 + 2048)
    if save_s:
        b += 16 * E_s
    if save_n:
        b += 16 * E_n
    return b, 3360 * E_s + 2320 * E_n + 7168 * N


# newest committed counter stamp of the K1 forward kernel (rocprofv3 --pmc passes of tools/k1_run.py, tools/k1_counters_json.py)
K1_COUNTERS = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_k1_hetero_counters.json") for r in (6, 5, 4)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r04_k1_hetero_counters.json"))
K1_COUNTERS_REL = os.path.relpath(K1_COUNTERS, ROOT)
K1_ENV_ROCPROF = os.path.join(ROOT, "profiles", "r06_k1_env_standalone_by_grid.txt")


def k1_env_rocprof(alg_bytes):
    """SURVEY 8(d) prices the graded kernel on its rocprofv3 kernel-trace duration: the committed trace of 50 back-to-back D-env rollout
    launches (tools/k1_run.py --dist env --reps 50 under rocprofv3 --kernel-trace; per-(kernel, grid) summary of tools/rocprof_by_grid.py)
    -> kernel-only p50 duration, without the launch boundary an event pair carries.  None when the file is absent."""
    if not os.path.exists(K1_ENV_ROCPROF):
        return None
    for ln in open(K1_ENV_ROCPROF):
        f = ln.split()
        if "gatv2_hetero_fwd_kernel" in ln and len(f) > 8 and f[0].isdigit() and int(f[0]) >= 40:
            p50, mn = float(f[3]), float(f[4])
            return {"source": os.path.relpath(K1_ENV_ROCPROF, ROOT) + " (committed; NOT collected inside this run - another box of the pool)",
                    "launches": int(f[0]), "p50_us": p50, "min_us": mn, "alg_bytes_per_launch": alg_bytes,
                    "achieved": alg_bytes / p50 / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / p50 / 1e3 / HBM_PEAK_GBS}
    return None



def measured_traffic(dist_name, n_inf, n_tr, B, n, M):
    """HBM bytes per fused K1 launch from the committed rocprofv3 PMC passes (K1_COUNTERS; method
    and gfx950 correction are documented there), averaged over the inference / training launches of a step like
    ``achieved``.  (None, None) when the workload is not the profiled one."""
    if not os.path.exists(K1_COUNTERS) or (B, n, M) != (4096, 8, 80):
        return None, None
    t = json.load(open(K1_COUNTERS)).get(dist_name)
    if not t:
        return None, None
    by = {k: (2.0 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024.0 for k, v in t.items()}
    src = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/k1_run.py on this workload, stamped in "
           + K1_COUNTERS_REL + " (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md); not "
           "collected inside this run")
    return (n_inf * by["inference"] + n_tr * by["training"]) / (n_inf + n_tr), src


def pipe_bound(dist_name, avg_launch_ms_rollout, clock_mhz, B, n, M):
    """What bounds a ROLLOUT launch of K1 on the instruction pipes, from the committed counter passes (SQ_INSTS_MFMA,
    SQ_INSTS_VALU of one inference launch at C3 size): a SIMD issues a v_mfma_f32_16x16x32_bf16 every ~17 cycles and a VALU
    instruction every 4 (MI355X_MICROARCH.md, per-instruction constants; DESIGN.md section 5: the two do not overlap inside one
    wavefront, and two wavefronts share a SIMD), so the issue floor of a launch is max(N_mfma x 17, N_valu x 4) cycles spread
    over 1024 SIMDs at the SUSTAINED shader clock sampled in this run.  frac_of_pipe_bound = that floor / measured time."""
    if not os.path.exists(K1_COUNTERS) or (B, n, M) != (4096, 8, 80) or not avg_launch_ms_rollout:
        return None
    t = json.load(open(K1_COUNTERS)).get(dist_name, {}).get("inference")
    if not t or "SQ_INSTS_VALU" not in t:
        return None
    clk = (clock_mhz or 2400.0) * 1e6
    cyc_mfma, cyc_valu = 17.0 * t["SQ_INSTS_MFMA"], 4.0 * t["SQ_INSTS_VALU"]
    floor_s = max(cyc_mfma, cyc_valu) / (1024.0 * clk)
    sum_s = (cyc_mfma + cyc_valu) / (1024.0 * clk)
    N = B * n
    return {"launch_class": "rollout (inference launch over B n destinations)",
            "valu_insts_per_launch": t["SQ_INSTS_VALU"], "mfma_insts_per_launch": t["SQ_INSTS_MFMA"],
            "salu_insts_per_launch": t.get("SQ_INSTS_SALU"), "valu_per_destination": t["SQ_INSTS_VALU"] / N,
            "salu_per_destination": (t.get("SQ_INSTS_SALU") or 0.0) / N,
            "clock_mhz_used": clk / 1e6, "clock_source": "sampled in this run" if clock_mhz else "nominal 2400 MHz (no sample)",
            "issue_floor_us_max_of_pipes": 1e6 * floor_s, "issue_floor_us_sum_of_pipes": 1e6 * sum_s,
            "measured_us": 1e3 * avg_launch_ms_rollout,
            "frac_of_pipe_bound": 1e6 * floor_s / (1e3 * avg_launch_ms_rollout),
            "frac_of_additive_pipe_bound": 1e6 * sum_s / (1e3 * avg_launch_ms_rollout),
            "counter_source": K1_COUNTERS_REL + " (rocprofv3 --pmc passes of tools/k1_run.py, not collected inside this run)"}


class ClockSampler:
    """Shader clock while the timed region runs: a thread reads hwmon's freq1_input (sclk, Hz) of THIS device every 20 ms
    (sysfs, no subprocess on the launch thread's core).  The box's sysfs lists every GPU of the node; the card is matched by
    the PCI address PyTorch reports for the device.  None when no card matches."""

    def __init__(self, index=0):
        import glob
        import threading
        self.path = None
        try:
            pr = th.cuda.get_device_properties(index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
                if os.path.basename(os.path.realpath(os.path.join(card, "device"))) == want:
                    f = sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*", "freq1_input")))
                    self.path = f[0] if f else None
        except Exception:   # noqa: BLE001 - an optional diagnostic never breaks the bench
            self.path = None
        self.samples, self._stop, self._thr, self._threading = [], False, None, threading

    def _read(self):
        try:
            return float(open(self.path).read().strip()) / 1e6
        except Exception:   # noqa: BLE001
            return None

    def __enter__(self):
        if self.path is not None:
            def loop():
                while not self._stop:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.02)
            self._thr = self._threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)
        return False

    def summary(self):
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"mean_mhz": sum(s) / len(s), "min_mhz": s[0], "max_mhz": s[-1], "samples": len(s),
                "source": self.path + " (sclk of this device, 20 ms period, timed region only)"}


def cpu_baseline(n, M, dist_name, T_s=2, B_fixed=64):
    """The oracle timed on the host cores (SURVEY 8d protocol) on a bounded, FIXED sample of the same cycle - T_s rollout
    forwards + one update (2 T_s + 1 forwards + BPTT backward) on B_fixed = 64 env graphs of the SAME degree distribution as
    the GPU leg, the same sample every round (rounds 1-3 calibrated the sample size per run: 512 / 128 / 64 graphs, and the
    figure moved by +-30 %): per thread count 2 warm-ups + median of 5 timed cycles (3 at one thread), threads in {1, 8, 32}
    (hosts with more than 64 threads additionally get a bounded single-forward probe at all threads, see below); ``value`` is
    the best multi-thread figure (``cores`` = its thread count), the 1-thread figure and the whole sweep are reported next
    to it."""
    import statistics

    from oracle import restatement as R
    from uav_bs_ctrl_amd import GnnAgent
    ncpu = os.cpu_count() or 1
    cfg = dict(enc="gnn", c="tarmac", n_heads=4, key_size=16, msg_size=64, n_rounds=1, dueling=False)
    net = GnnAgent(dict(agent=2, ubs=2, gt=4), 9, exp3_args("cpu"))
    p = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    p_t = {k: v.detach().clone() for k, v in p.items()}
    gen = th.Generator().manual_seed(5)

    def make(B_s):
        N = B_s * n
        En = N * (n - 1)
        base = (th.arange(N) // n * n).repeat_interleave(n)

        def graph():
            if dist_name == "dense":
                d_seen = th.full((N,), M, dtype=th.int64)
            else:   # D-env: no GT in sight w.p. 0.94, else U{1..0.65 M}
                hi = max(1, int(0.65 * M))
                d_seen = th.where(th.rand(N, generator=gen) < 0.94, th.zeros(N, dtype=th.int64),
                                  th.randint(1, hi + 1, (N,), generator=gen))
            so = th.zeros(N + 1, dtype=th.int32)
            so[1:] = th.cumsum(d_seen, 0)
            Es = int(so[-1])
            return dict(x_a=th.rand(N, 2, generator=gen), x_gt=th.rand(Es, 4, generator=gen) * 2 - 1, seen_off=so,
                        x_ubs=th.rand(En, 2, generator=gen) * 2 - 1,
                        near_off=th.arange(0, En + 1, n - 1, dtype=th.int32),
                        talk_off=th.arange(0, N * n + 1, n, dtype=th.int32),
                        talk_src=(base + th.arange(n).repeat(N)).to(th.int32))
        obs = [graph() for _ in range(T_s + 1)]
        acts = th.randint(9, (T_s, N, 1), generator=gen)
        rews, dones = th.rand(T_s, B_s, n, generator=gen), th.zeros(T_s, B_s, 1)

        def cycle():
            h = th.zeros(N, 256)
            with th.no_grad():
                for t in range(T_s):
                    _, h = R.gnn_agent_forward(obs[t], h, p, cfg)
            loss, _, _ = R.madrqn_loss(obs, th.zeros(N, 256), th.zeros(N, 256), acts, rews, dones, p, p_t, cfg, 0.99)
            th.autograd.grad(loss, list(p.values()))
        return cycle

    def timed(cycle):
        t0 = time.perf_counter()
        cycle()
        return time.perf_counter() - t0

    counts = sorted({c for c in (1, 8, 32) if c <= ncpu})
    sweep = {}
    all_core_probe = None
    if ncpu > 64:
        # All host threads: PyTorch's intra-op pool over hundreds of threads turns every one of the ~2000 small ops of a
        # cycle into a many-millisecond rendezvous (measured on this pool's 256-thread hosts: 69-89 s for ONE cycle on 2 env
        # graphs, 1000x slower than 8 threads), so a full cycle cannot be afforded inside a bounded baseline.  A bounded
        # probe documents it instead: one no-grad forward on one env graph at all threads vs at 32.
        def one_forward(threads):
            th.set_num_threads(threads)
            N = n
            gsmall = dict(x_a=th.rand(N, 2, generator=gen), x_gt=th.rand(N * 4, 4, generator=gen),
                          seen_off=th.arange(0, N * 4 + 1, 4, dtype=th.int32), x_ubs=th.rand(N * (n - 1), 2, generator=gen),
                          near_off=th.arange(0, N * (n - 1) + 1, n - 1, dtype=th.int32),
                          talk_off=th.arange(0, N * n + 1, n, dtype=th.int32), talk_src=th.arange(n).repeat(N).to(th.int32))
            with th.no_grad():
                R.gnn_agent_forward(gsmall, th.zeros(N, 256), p, cfg)      # warm-up
                t0 = time.perf_counter()
                R.gnn_agent_forward(gsmall, th.zeros(N, 256), p, cfg)
            return time.perf_counter() - t0
        t_all, t_32 = one_forward(ncpu), one_forward(32)
        all_core_probe = dict(threads=ncpu, sec_one_forward_one_env_graph=t_all, sec_same_at_32_threads=t_32,
                              slowdown_vs_32_threads=t_all / max(t_32, 1e-9))
    for threads in counts:
        th.set_num_threads(threads)
        B_s = B_fixed
        c = make(B_s)
        timed(c)                                            # warm-up 1 (thread pool, allocator)
        timed(c)                                            # warm-up 2
        ts = [timed(c) for _ in range(3 if threads == 1 else 5)]
        med = statistics.median(ts)
        sweep[str(threads)] = dict(env_steps_per_s=B_s * T_s / med, sec_per_cycle_median=med, env_graphs=B_s,
                                   sec_min=min(ts), sec_max=max(ts))
    multi = {k: v for k, v in sweep.items() if int(k) > 1} or sweep
    best = max(multi, key=lambda k: multi[k]["env_steps_per_s"])
    return dict(value=sweep[best]["env_steps_per_s"], unit="env-steps/s", cores=int(best), kind="port",
                host_threads_available=ncpu, one_thread=sweep["1"]["env_steps_per_s"], thread_sweep=sweep,
                all_core_probe=all_core_probe,
                protocol=f"fixed sample of {B_fixed} env graphs; per thread count in {{1, 8, 32}}: 2 warm-up cycles, median of 5 "
                         "timed cycles (3 at one thread); value = best multi-thread count",
                torch_parallel_info=th.__config__.parallel_info().split("\n")[0],
                sample=f"oracle/restatement.py (PyTorch CPU fp32): cycles of {T_s} rollout forwards + 1 update "
                       f"({2 * T_s + 1} forwards + BPTT backward) on {sweep[best]['env_graphs']} env graphs of {n}x{M}, "
                       f"D-{dist_name} degrees (same distribution as the GPU leg); the reference's real DGL-CPU path is "
                       f"not installable on either box, so this is the build's CPU restatement (kind 'port')",
                sec_per_cycle=sweep[best]["sec_per_cycle_median"])


def end_to_end(learner, a, device, mode="random", cycles=3):
    """SURVEY 8f rows f1 + f2 + f3 inside the timed region (reported NEXT TO the headline, whose graphs are synthetic as
    BASELINE.json asks): B environments of the batched device simulator (csrc/env_sim.hip; exp3 'DenseHotSpot' physics,
    maps.py:82-111, at n x M) are rolled out for T steps - simulator step, device graph construction, act, replay push
    per step - then ONE update consumes the B stored sequences (graphs of all T+1 steps rebuilt from the replay's padded
    tensors).  env-steps/s = B T / cycle time.  Two operating points of the `seen` relation:
      mode "random"  - epsilon = 1, the uniformly random policy the reference starts training with (run.py:59-60,
                       learner.py:73-80), UBSs placed on the grid as maps.py:96-111 does: sparse `seen`;
      mode "hotspot" - UBSs placed ON the GT hotspot and held there (the policy forward and the action selection still
                       run, the simulator is stepped with the hover action): dense `seen`, what a trained policy's
                       rollouts look like.
    (An untrained network's near-greedy argmax, epsilon = 0.05, walks every UBS into a wall: visibility 3e-4, measured in
    round 2 - a zero-degree workload that says nothing.)"""
    from uav_bs_ctrl_amd import from_padded_obs
    from uav_bs_ctrl_amd.replay import SequenceReplay
    from uav_bs_ctrl_amd.sim import BatchedUbsCoverageEnv, MapParams
    B, n, M, T = a.B, a.n, a.M, a.T
    mp = MapParams(n_ubs=n, n_gts=M, n_rbs=5, range_pos=6000.0, episode_limit=T, dt=40.0, r_cov=100.0, r_sns=400.0,
                   vels=(5.0, 10.0), n_dirs=4, reward_scale_rate=10.0)
    env = BatchedUbsCoverageEnv(mp, B, device)
    rb = SequenceReplay(capacity=B, max_seq_len=T, n_agents=n, n_gts=M, hidden_size=256, n_envs=B, state_dim=0,
                        r_comm=mp.r_comm, device=device)
    gen = th.Generator(device=device)
    gen.manual_seed(99)
    STATIC = os.environ.get("UAVGNN_E2E_STATIC", "1") == "1"   # per-step graphs without the host round trip for edge totals
    # UAVGNN_E2E_GRAPH=1: the rollout step (device graph construction + policy forward + selection) as ONE hipGraph replay
    # from fixed-address observation buffers (uav_bs_ctrl_amd/graphs.py).  At B = 4096 the eager step is GPU-bound and the
    # replay buys nothing (measured round 3: 129.6 vs 128.8 ms per cycle); it pays at small B (profiles/r03_small_batch.txt).
    ga = None
    if os.environ.get("UAVGNN_E2E_GRAPH", "0") == "1":
        from uav_bs_ctrl_amd.graphs import GraphedAct
        ga = GraphedAct(learner, B, n, M, mp.r_comm)

    def positions():   # hotspot of M/5 groups of 5 GTs on a 200 m grid, UBSs on grid points (maps.py:96-111)
        grid = 200.0
        spot = th.randint(0, 26, (B, 1, 2), device=device, generator=gen).double() * grid
        grp = spot + th.randint(0, 4, (B, M // 5, 2), device=device, generator=gen).double() * grid
        gts = grp.repeat_interleave(5, 1) + 100.0 * (th.rand(B, M, 2, device=device, generator=gen, dtype=th.float64) - 0.5)
        if mode == "hotspot":   # inside the 800 m x 800 m hotspot, at least 2 safe distances apart (distinct 100 m cells)
            cell = th.rand(B, 64, device=device, generator=gen).argsort(1)[:, :n]                             # [B, n] distinct cells
            ubs = spot + th.stack((cell % 8, cell // 8), -1).double() * 100.0 + 50.0
        else:
            ubs = th.randint(0, 30, (B, n, 2), device=device, generator=gen).double() * grid
        return ubs.clamp(0, mp.range_pos), gts.clamp(0, mp.range_pos).float()


    vis = []

    def cycle():
        ubs, gts = positions()
        o = env.reset(ubs, gts, generator=gen)
        h = learner.init_hidden(B)
        for t in range(T):
            rb.stage_obs(dict(gt=o["gt"], ubs=o["ubs"], agent=o["agent"], d_u2u=o["d_u2u"], h=h.view(B, n, -1)))
            eps = 1.0 if mode == "random" else 0.05
            if ga is not None:
                acts, h2 = ga(o["gt"], o["ubs"], o["agent"], o["d_u2u"], h, eps)
            else:
                g = from_padded_obs(o["gt"], o["ubs"], o["agent"], o["d_u2u"], r_comm=mp.r_comm, static=STATIC)
                acts, h2 = learner.act(g, h, eps)
            if mode == "hotspot":
                acts = th.zeros_like(acts)        # hover: the UBSs stay on the hotspot
            o, rew, done, info = env.step(acts)   # overwrites the observation buffers in place: they are in the replay already
            vis.append(o["gt"][..., 0].mean())
            # learner.cache (learner.py:82-92): done muted by the time-limit mask, next_h zeroed by the raw done flag; the
            # observation half of the transition was staged before the step, so only the next_* fields travel here
            learner.cache(rb, {}, h, None, acts, rew.float(), dict(gt=o["gt"], ubs=o["ubs"], agent=o["agent"], d_u2u=o["d_u2u"]),
                          h2, None, done, info["BadMask"], staged=True)
            h = h2
        m = rb.mem                                              # all B sequences, time-major padded tensors
        tm = {k: m[k].transpose(0, 1).contiguous() for k in ("gt", "ubs", "agent", "d_u2u")}
        obs = [from_padded_obs(tm["gt"][t], tm["ubs"][t], tm["agent"][t], tm["d_u2u"][t], r_comm=mp.r_comm, static=STATIC)
               for t in range(T + 1)]
        flat = lambda x, lo: x[lo:].reshape((-1,) + x.shape[2:])  # noqa: E731
        batch = dict(obs=obs, obs_all=from_padded_obs(flat(tm["gt"], 0), flat(tm["ubs"], 0), flat(tm["agent"], 0)),
                     obs_all_next=from_padded_obs(flat(tm["gt"], 1), flat(tm["ubs"], 1), flat(tm["agent"], 1)),
                     h0=m["h"][:, 0].reshape(B * n, -1), h1=m["h"][:, 1].reshape(B * n, -1),
                     acts=m["act"].permute(1, 0, 2).reshape(T, B * n, 1), rews=m["rew"].permute(1, 0, 2).contiguous(),
                     dones=m["done"].permute(1, 0, 2).contiguous())
        return learner.update(batch)

    gc.collect()
    th.cuda.empty_cache()        # every leg starts from a clean caching allocator: the blocks the previous legs left behind (the
    cycle()                      # rho leg's 32 chunks, the staging buffers) cost the hotspot leg 28 ms per cycle otherwise
    cycle()                      # (two warm-up cycles: the first one regrows the allocator's pools)
    th.cuda.synchronize()
    vis.clear()
    # every cycle timed on its own, the MEDIAN reported: single cycles of this leg were seen at 3-5x the others (365 ms vs 123
    # over two runs of the same build on one box - pool regrowth / host hiccups of a leg that steps a simulator from Python); the
    # mean of two cycles made the figure a coin toss
    per_cycle = []
    for _ in range(cycles):
        t0 = time.perf_counter()
        out = cycle()
        th.cuda.synchronize()
        per_cycle.append(time.perf_counter() - t0)
    dt = sorted(per_cycle)[len(per_cycle) // 2]
    served = float((env.out["gt_ubs"] >= 0).float().mean())
    seen = float(th.stack(vis).mean())            # over every step of the timed cycles, not just the last one
    return dict(value=B * T / dt, unit="env-steps/s", ms_per_cycle=1e3 * dt, cycles=cycles,
                ms_per_cycle_all=[round(1e3 * t, 2) for t in per_cycle], statistic="median of the timed cycles", loss=float(out["LossQ"]),
                mode=mode, policy=("uniformly random actions (epsilon = 1)" if mode == "random" else
                                   "policy forward + selection run, simulator stepped with the hover action"),
                includes="batched device simulator (f3) + device graph construction (f1) + tensor replay (f2) + act + update",
                rollout_step=("one hipGraph replay (GraphedAct: graph construction + policy forward + selection)" if ga is not None
                              else "eager launches"),
                physics=f"DenseHotSpot-style map at {n} x {M}: 5 RBs, r_cov 100 m, r_sns 400 m, range 6 km, dt 40 s",
                mean_gt_visibility=seen, mean_d_seen=seen * M, mean_gt_served=served)


_SPAN_FLOOR = None


def event_span_floor_us():
    """What a HIP-event pair reports around (a) nothing and (b) the smallest kernel (one 4-byte fill) on this box and stream: the
    bias of `avg_launch_ms` - an event span contains the dispatch gap in front of the kernel it brackets, a rocprofv3 kernel
    duration does not (profiles/*_by_grid.txt: p50 of the K1 launches vs `by_launch_class.rollout.avg_launch_ms`).  `achieved` is
    computed from the raw spans (pessimistic by this much per launch).  Median of 25."""
    global _SPAN_FLOOR
    if _SPAN_FLOOR is None:
        x = th.zeros(1, device="cuda")
        res = {}
        for name, fn in (("empty", lambda: None), ("tiny_kernel", lambda: x.fill_(1.0))):
            ts = []
            for _ in range(28):
                e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(1e3 * e0.elapsed_time(e1))
            res[name] = round(sorted(ts[3:])[len(ts[3:]) // 2], 2)
        _SPAN_FLOOR = res
    return _SPAN_FLOOR


def mfma_calibration(device, n=8192, reps=120):
    """TFLOP/s of the vendor's bf16 GEMM (hipBLASLt through torch.mm) on an n^3 problem, ~0.1 s of back-to-back launches (long
    enough for the power management to settle): the practical matrix-core roof of this box, for scale.  None if anything fails."""
    try:
        a = th.randn(n, n, device=device, dtype=th.bfloat16)
        b = th.randn(n, n, device=device, dtype=th.bfloat16)
        for _ in range(5):
            th.mm(a, b)
        th.cuda.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            th.mm(a, b)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
        return {"kernel": f"hipBLASLt bf16 GEMM {n}^3 (torch.mm), {reps} back-to-back launches", "ms_per_launch": ms, "tflops": tf,
                "frac_of_nominal_peak": tf / BF16_PEAK_TFLOPS}
    except Exception:   # noqa: BLE001 - an optional diagnostic never breaks the bench
        return None


GRAPHED_CYCLE_ROWS = 8192     # agent rows per GPU at or below which --graphed-cycle auto replays the cycle from a hipGraph


def k1_roofline(k, a, dist_name, clock_mhz):
    """``roofline`` object of the fused K1 forward from the HIP-event spans `k` of a timed region (every launch of it:
    rollout launches over N_a destinations, the two time-batched encoder launches of an update over (T+1) N_a / T N_a
    destinations): achieved = sum(algorithmic bytes or flops) / sum(event durations)."""
    work = [(Es, En, N, ss, sn) for (Es, En, N, ss, sn) in k["work"]]
    tot_b = sum(alg_k1_fwd(*w)[0] for w in work)
    tot_f = sum(alg_k1_fwd(*w)[1] for w in work)
    sec = k["total_ms"] * 1e-3
    ach = tot_b / sec / 1e9
    by_class = {}
    floor_ms = event_span_floor_us()["empty"] * 1e-3
    for cls, sel in (("rollout", lambda N: N <= a.B * a.n), ("update_time_batched", lambda N: N > a.B * a.n)):
        ms_c = [m for m, w in zip(k["ms"], work) if sel(w[2])]
        by_c = [alg_k1_fwd(*w)[0] for w in work if sel(w[2])]
        fl_c = [alg_k1_fwd(*w)[1] for w in work if sel(w[2])]
        if ms_c:
            g = sum(by_c) / (sum(ms_c) * 1e-3) / 1e9
            tf = sum(fl_c) / (sum(ms_c) * 1e-3) / 1e12
            by_class[cls] = {"launches": len(ms_c), "avg_launch_ms": sum(ms_c) / len(ms_c), "achieved": g,
                             "frac": g / HBM_PEAK_GBS, "hbm_frac": g / HBM_PEAK_GBS, "tflops": tf,
                             "mfma_fp32_frac": tf / FP32_PEAK_TFLOPS}
            # the same with the span of an EMPTY event pair taken off every launch (what rocprofv3's kernel durations show; the
            # raw figures above are the ones `achieved` / `frac` of the object are built from)
            net = [max(m - floor_ms, 1e-6) for m in ms_c]
            by_class[cls]["avg_launch_ms_net_of_event_floor"] = sum(net) / len(net)
            by_class[cls]["hbm_frac_net_of_event_floor"] = sum(by_c) / (sum(net) * 1e-3) / 1e9 / HBM_PEAK_GBS
    n_units = sum(w[2] for w in work) / (a.B * a.n)                # launches in units of one env-step batch
    n_tr = sum(w[2] for w in work if w[3]) / (a.B * a.n)
    traffic, traffic_src = measured_traffic(dist_name, n_units - n_tr, n_tr, a.B, a.n, a.M)
    tfl = tot_f / sec / 1e12
    # The roof that binds is the lower one at this launch mix's arithmetic intensity (classic roofline):
    # AI = algorithmic FLOP / algorithmic byte vs the machine balance fp32-MFMA peak / HBM peak (~19.7 FLOP/B).
    ai = tot_f / max(tot_b, 1)
    balance = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    hbm = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    mfma = {"achieved": tfl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP32_PEAK_TFLOPS}
    bound = "mfma" if ai > balance else "hbm"
    top = mfma if bound == "mfma" else hbm
    return {"bound": bound,
            "kernel": "gatv2_hetero_fwd_kernel (K1 forward, `seen` + `near` relations in one launch)",
            "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
            "frac": top["frac"],
            "traffic": None if traffic is None else traffic * n_units / k["count"],
            "traffic_source": traffic_src,
            "arithmetic_intensity": ai, "machine_balance": balance,
            "hbm": hbm, "mfma_fp32": mfma,
            "avg_launch_ms": k["avg_ms"], "launches": k["count"],
            "alg_bytes_per_launch": tot_b / k["count"], "alg_flops_per_launch": tot_f / k["count"],
            "ms_per_env_step_batch": k["total_ms"] / n_units,
            "by_launch_class": by_class,
            "pipe_bound": pipe_bound(dist_name, by_class.get("rollout", {}).get("avg_launch_ms"), clock_mhz, a.B, a.n, a.M),
            "event_span_floor_us": event_span_floor_us(),
            "note": ("D-dense: AI ~ 86 FLOP/B (SURVEY 8d) is 4x the machine balance, so the roof SURVEY 8d prices the kernel against "
                     "is the fp32 MFMA / vector peak (157.3 TF); even there it could move only ~23 % of the HBM peak.  `frac` is "
                     "that throughput ratio and NOT a bound the kernel lives under: the score GEMM (2048 of the 3360 FLOP per "
                     "`seen` edge) and the epilogue products of `near` run on the bf16 matrix cores as six exact bf16 products per "
                     "fp32 product.  What does bound a launch is instruction issue - `pipe_bound` (counter passes + the clock "
                     "sampled in this run)"
                     if bound == "mfma" else
                     "D-env (94 % of agents see no GT): AI below the machine balance, the launch mix is bound by HBM "
                     "(output-row writes: 2 KB per agent, 67 MB per rollout launch; a pure streaming write of them runs at "
                     "5.2-6.2 TB/s on this part, profiles/r04_k1_standalone.txt)")}


def k1_env_standalone(learner, g, a, reps=50, rounds=15):
    """The graded kernel ALONE and back to back, inside the driver's run: `reps` launches of the fused K1 forward (prepared
    parameter image, `uavgnn_gatv2_hetero_fwd_image`: the launch every rollout step of `learner.act` makes) on ONE D-env rollout
    batch, recorded into a hipGraph and replayed between ONE HIP-event pair - so the span holds `reps` kernels + their `reps - 1`
    dependent-launch boundaries and ONE event floor (4.5-6.3 us / `reps`), instead of one floor per launch as in
    `by_launch_class.rollout` (events around every launch of the cycle).  What tools/ubench/k1_env_bench.hip measures on the
    builder's side (profiles/r0*_k1_standalone.txt), here through the product's own op (`ops.hetero_gatv2`) and C-ABI entry.
    Median of `rounds` replays enqueued back to back after 40 warm-up replays (the fastest span is listed next to it); algorithmic bytes as `alg_k1_fwd`."""
    from uav_bs_ctrl_amd import ops
    enc = learner.policy_net.enc
    x_a = g.agent_feat()
    rels = []
    for et in ("seen", "near"):
        x_src, off = g.relation_segments(et)
        rels.append((x_src, off, g.relation_order(et), enc.f_conv[et]))
    E_s, E_n, N = rels[0][0].shape[0], rels[1][0].shape[0], x_a.shape[0]
    store = {}
    from uav_bs_ctrl_amd.graphs import _capture
    graph = th.cuda.CUDAGraph()
    with th.no_grad(), ops.frozen_weights(store):
        ops.hetero_gatv2(x_a, enc._n_heads, rels)           # builds the image + any derived index outside the capture
        th.cuda.synchronize()
        with _capture(graph):      # thread-local capture mode when a process group (and its watchdog thread) is alive
            out = None
            for _ in range(reps):
                del out     # the previous launch's [N, 2H] rows go back to the pool first: every launch writes the SAME 67 MB, as the
                out = ops.hetero_gatv2(x_a, enc._n_heads, rels)   # rollout does (the allocator hands act() the block f_aggr just freed)
    took_image = any(k[0] == "k1img" for k in store)
    for _ in range(40):          # ~40 ms of back-to-back launches: the spans below start from a settled clock (the first spans after an
        graph.replay()           # idle gap read 10-15 % long: 22.1, 20.9, 19.4 ... 18.6 us)
    ms = []
    for i in range(rounds):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        ms.append((e0, e1))
    th.cuda.synchronize()
    ms = [e0.elapsed_time(e1) for e0, e1 in ms]
    del graph, out
    med = sorted(ms)[len(ms) // 2]
    by, fl = alg_k1_fwd(E_s, E_n, N, False, False)
    us = 1e3 * med / reps
    return {"launches_per_span": reps, "spans": rounds, "us_per_launch": us, "us_per_launch_all": [round(1e3 * m / reps, 3) for m in ms],
            "alg_bytes_per_launch": by, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "tflops": fl / (us * 1e-6) / 1e12,
            "us_per_launch_best_span": 1e3 * min(ms) / reps, "frac_best_span": by / (1e3 * min(ms) / reps * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "prepared_image": took_image, "destinations": N, "seen_edges": E_s, "near_edges": E_n,
            "how": f"{reps} launches recorded into a hipGraph and replayed between ONE HIP-event pair (launch boundaries included, "
                   "one event floor per span), no-grad rollout launch through ops.hetero_gatv2 -> uavgnn_gatv2_hetero_fwd_image"}


_REAL_STDOUT = None


def emit(line: str) -> None:
    """The ONE JSON line of the run, to the process's original stdout (see main)."""
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    print(line, file=out, flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(a):
    """``python3 bench.py --gpus N`` started PLAIN (no RANK / WORLD_SIZE in the environment - the way the driver starts the N = 1
    run) starts its N ranks itself: it re-executes this file under ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`` - one process per GPU over RCCL, exactly the
    documented launch form, which keeps working unchanged (a process that finds RANK in its environment is a rank and never
    re-launches).  The reference's counterpart is ``mpi_fork`` (utils/mpi_tools.py:6-36: re-exec under ``mpirun -np n``).
    Returns the exit code of the launcher; rank 0's ONE JSON line goes to this process's stdout untouched."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")                # torch.distributed.run would set 1 and say so on stderr
    env["UAVGNN_BENCH_SELF_LAUNCHED"] = "1"
    print(f"[bench] --gpus {a.gpus} without RANK/WORLD_SIZE: launching {a.gpus} ranks via torch.distributed.run", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def rccl_transport_summary(path):
    """What RCCL said about its transports in this rank's NCCL_DEBUG=INFO log (set by this file before the communicator is
    created when the caller did not choose a debug level): `Channel .. via <transport>` lines counted per transport, plus the
    lines that name the topology / xGMI.  None when the log is missing (gloo test hook, or a caller-owned NCCL_DEBUG)."""
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError:
        return None
    via, topo = {}, []
    for ln in lines:
        m = re.search(r"\bvia\s+([A-Za-z0-9_/+ -]+?)(?:\s*$|\s+comm|\s*\[)", ln)
        if m and "Channel" in ln:
            via[m.group(1).strip()] = via.get(m.group(1).strip(), 0) + 1
        if re.search(r"xgmi|XGMI|Connected all|nRanks|topology|Trees|Rings? ", ln) and len(topo) < 12:
            topo.append(ln.split("NCCL INFO", 1)[-1].strip()[:200])
    return {"channel_transports": via, "lines": topo, "log_lines": len(lines), "source": "NCCL_DEBUG=INFO log of rank 0"}


def peer_access_summary(world):
    """Which of the node's first `world` devices can address each other's memory directly (hipDeviceCanAccessPeer through torch):
    what RCCL's P2P / xGMI transports need.  Cheap, silent, and independent of any debug log."""
    try:
        n = min(world, th.cuda.device_count())
        ok = [[bool(i == j or th.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
        return {"devices": n, "all_pairs": all(all(r) for r in ok), "pairs_without_peer_access": [[i, j] for i in range(n) for j in range(n) if not ok[i][j]]}
    except Exception as e:   # noqa: BLE001 - a diagnostic never breaks the bench line
        return {"error": repr(e)}


def dry_launch_main(a, world, rank):
    """Test hook UAVGNN_BENCH_DRY=1 (tests/test_bench_launch.py, CPU container: no GPU): everything of the launch contract that
    does not need a device - rank discovery, process-group rendezvous (gloo), barrier-bracketed timing of K trivial steps,
    MAX over ranks, ONE JSON line from rank 0 - so that both launch forms (plain ``python3 bench.py --gpus N`` and
    ``torch.distributed.run``) are exercised where the kernels cannot run.  The line says ``dry_run``; it is not a measurement."""
    if world > 1:
        dist.init_process_group("gloo")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    el = th.tensor([time.perf_counter() - t0], dtype=th.float64)
    ck = th.tensor([1.0, 2.0], dtype=th.float64)
    same = True
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(th.equal(lo, hi))
    if rank == 0:
        emit(json.dumps({"metric": metric_name(a), "value": 0.0, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": 1e3 * float(el) / max(a.steps, 1), "dry_run": True,
                          "replicas_identical": same, "rccl_ranks": world,
                          "self_launched": os.environ.get("UAVGNN_BENCH_SELF_LAUNCHED") == "1",
                          "config": {"global_batch": world * a.B, "parallelism": f"dp{world}"}}))
    if world > 1:
        dist.destroy_process_group()


def metric_name(a):
    """BASELINE.json's metric with the sizes of THIS run (the headline run is 8 UBS x 80 GT)."""
    return f"env-steps/sec (MADRQN, {a.n} UBS x {a.M} GT)"



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    # two warm-up cycles by default: the second cycle runs while the first one's outputs are still alive and needs one block more - with ONE
    # warm-up cycle that device allocation (a hipMalloc: 59 ms on boxes whose VRAM other processes had just used,
    # profiles/r06_final_bench_dense_*_host_stall.json) falls into the timed region (`device_allocs_by_step` in the line)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--B", type=int, default=4096, help="env graphs per GPU")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--M", type=int, default=80)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--dist", default="dense", choices=["dense", "env"])
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic graphs cycled through the T+1 steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the simulator-inclusive figure (rows f1+f2+f3)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even at world size 1 (smoke test)")
    ap.add_argument("--rho", type=int, default=32, help="replay ratio of the extra `replay_ratio_leg` (headline: 1)")
    ap.add_argument("--no-rho-leg", action="store_true", help="skip the replay-ratio leg (one cycle with rho chunks per update)")
    ap.add_argument("--no-env-leg", action="store_true", help="skip the short D-env leg (`roofline_env`: the HBM-bound regime of K1)")
    ap.add_argument("--env-steps", type=int, default=5, help="timed steps of the D-env leg")
    ap.add_argument("--graphed-cycle", default="auto", choices=["auto", "on", "off"],
                    help="EXTRA leg: the cycle replayed from one hipGraph (graphs.GraphedCycle) -> `value_graphed`; auto: below %d agent "
                         "rows per GPU, single process.  `value` is always the eager cycle" % GRAPHED_CYCLE_ROWS)
    ap.add_argument("--no-fp32-leg", action="store_true",
                    help="skip the second timing with the bf16x3 kernels off (fp32-MFMA GRU cell, vendor fp32 GEMMs)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # plain `python3 bench.py --gpus N`: this process is the launcher, not a rank
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but this rank was started with WORLD_SIZE={world} (RANK={rank}); start it as "
                 f"`python3 bench.py --gpus {a.gpus}` (it launches its ranks itself) or as `python -m torch.distributed.run "
                 f"--nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {a.gpus} ...`")
    # ONE line on stdout, whatever the libraries print: the descriptor of the real stdout is kept for the JSON line and fd 1 is
    # pointed at stderr for everything else (the GPU box sets NCCL_DEBUG=VERSION, at which RCCL writes a five-line version banner
    # to the C-level stdout of every rank that creates a communicator).
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("UAVGNN_BENCH_DRY") == "1":
        return dry_launch_main(a, world, rank)
    # test hooks (tests/test_dp_gpu.py): all ranks on ONE device over gloo, so that the world-size-2 code path of this file
    # runs on a one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver: one rank per GPU over RCCL.
    backend = os.environ.get("UAVGNN_BENCH_BACKEND", "nccl")
    if os.environ.get("UAVGNN_BENCH_ONE_DEVICE") == "1":
        local = 0
    th.cuda.set_device(local)
    device = th.device("cuda", local)
    use_dist = world > 1 or (a.force_dist and "RANK" in os.environ)
    rccl_log = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            # The transport RCCL picked is read from the caller's OWN debug log when there is one (NCCL_DEBUG=INFO together with
            # NCCL_DEBUG_FILE=<path with %p / %h>).  The file never switches the log on by itself: at NCCL_DEBUG=INFO this RCCL
            # (2.26.6) prints a version banner to STDOUT whatever NCCL_DEBUG_FILE says - five lines in front of the ONE JSON
            # line a driver parses (seen with --force-dist on the GPU box).  Without a log the line carries the peer-access
            # matrix of the node instead (`peer_access`).
            dbg_file = os.environ.get("NCCL_DEBUG_FILE")
            if os.environ.get("NCCL_DEBUG", "").upper() == "INFO" and dbg_file:
                import socket
                rccl_log = dbg_file.replace("%p", str(os.getpid())).replace("%h", socket.gethostname())
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from uav_bs_ctrl_amd import enable_tuned_gemms, ops
    from uav_bs_ctrl_amd.learner import MultiAgentQLearner, params_checksum
    tuned = enable_tuned_gemms()    # recorded vendor-GEMM solutions (selection only; opt-in, process-wide)

    th.manual_seed(0)
    env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=a.n, episode_limit=a.T)
    learner = MultiAgentQLearner(env_info, exp3_args(str(device)))
    if use_dist and a.force_dist:
        learner.grads.force_collective = True    # world size 1: still push the flat gradient buffer through RCCL
    batch = make_sequence(a.B, a.n, a.M, a.T, a.dist, device, seed=1234 + rank, distinct=a.distinct)

    def step():
        # Every cycle sees NEWLY BUILT graphs: .fresh() shares the arrays but drops every derived index (K1 hand-out
        # order, talk-CSC transpose), so building them is inside the timed region - as with fresh observations from a
        # simulator or a new replay sample (the reference pays DGL's lazy CSR/CSC materialisation the same way).
        obs = [g.fresh() for g in batch["obs"]]
        fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
        h = learner.init_hidden(a.B)
        for t in range(a.T):
            _, h = learner.act(obs[t].fresh(), h, 0.05)
        return learner.update(fb)

    def barrier():
        if use_dist:
            dist.barrier()
        th.cuda.synchronize()

    # Small batches (C2: 4 096 agent rows) are launch-bound: the same cycle, captured once and replayed (every input of `step`
    # already sits at a fixed device address), is reported as an EXTRA leg (`graphed_cycle_leg` / `value_graphed`) behind the timed
    # region.  The headline `value` is always the eager cycle: a replay bakes the stored batch's topology into the graph, which a
    # loop fed fresh observations could only reproduce with capacity-sized (static) graphs.
    want_graphed = not use_dist and (a.graphed_cycle == "on" or (a.graphed_cycle == "auto" and a.B * a.n <= GRAPHED_CYCLE_ROWS))
    eager_step = step

    # Inside the timed region only the GRADED kernel (K1 forward) carries HIP events: timing every launch costs the launch thread
    # ~15 us per launch and makes the rollout phase host-bound (tools/launch_bound_probe.py: 21 -> 29 ms per 50 act forwards).
    # The other kernels are timed in ONE extra, fully instrumented cycle after the timed region (`instrumented_cycle`).
    GRADED = ("gatv2_hetero_fwd",)
    ops.KERNEL_TIMER.reset(enabled=True, only=GRADED)   # warm-up runs instrumented the same way
    # Python's cyclic GC is kept out of the timed region (a gen-2 pass over the autograd objects of a 101-forward
    # BPTT graph stalls the launch thread for 70-100 ms roughly every 8th step); memory is released by reference
    # counting as usual.  It is switched off IN FRONT of the warm-up cycles, not between them and the timed ones: a collection at that
    # point hands blocks back to the caching allocator that the cycles without GC then hold a little longer, and the second timed cycle
    # could make a device allocation (a hipMalloc inside the timed region: `device_allocs_by_step` in the line)
    gc.collect()
    gc.disable()
    out = None
    for _ in range(a.warmup):
        out = step()      # (kept, as the timed loop keeps it: the previous cycle's outputs are alive while the next one runs - with the result
    barrier()             # dropped here the SECOND timed cycle needed one block more than any warm-up cycle: the same hipMalloc)
    ops.KERNEL_TIMER.reset(enabled=True, only=GRADED)
    learner.grads.collective_events = []            # HIP events around the one collective of an update (when there is one)
    clock = ClockSampler(local)
    alloc0 = th.cuda.memory_stats(device).get("num_device_alloc", 0)
    with clock:
        t0 = time.perf_counter()
        marks, enq, allocs_by_step = [], [], []
        for _ in range(a.steps):
            out = step()
            ev = th.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
            enq.append(time.perf_counter() - t0)     # when the launch thread had ENQUEUED the step (no synchronisation)
            allocs_by_step.append(th.cuda.memory_stats(device).get("num_device_alloc", 0))
        barrier()
        elapsed = time.perf_counter() - t0
    alloc1 = th.cuda.memory_stats(device).get("num_device_alloc", 0)
    ktimes = ops.KERNEL_TIMER.summary()
    coll_ms = [e0.elapsed_time(e1) for e0, e1 in learner.grads.collective_events]
    learner.grads.collective_events = None
    clock_info = clock.summary()
    clock_mhz = clock_info["mean_mhz"] if clock_info else None
    ops.KERNEL_TIMER.reset(enabled=True)              # every span: one more cycle, on every rank (the update holds a collective)
    t_i = time.perf_counter()
    eager_step()
    barrier()
    instr_s = time.perf_counter() - t_i
    kfull = ops.KERNEL_TIMER.summary()
    gc.enable()
    ops.KERNEL_TIMER.enabled = False
    el = th.tensor([elapsed], device=device, dtype=th.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el)
    loss = float(out["LossQ"])
    ck = params_checksum(learner.policy_net)
    if use_dist:   # replicas must stay bit-identical
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(th.equal(lo, hi))
        assert replicas_identical, "parameter replicas diverged"
    else:
        replicas_identical = True

    # (behind the loss / parameter checksum of the TIMED steps: the leg's own updates move the parameters further)
    graphed_leg = None
    if want_graphed:
        from uav_bs_ctrl_amd.graphs import GraphedCycle
        ops.KERNEL_TIMER.reset(enabled=False)           # HIP events cannot be read back from inside a replayed graph
        h_row = learner.init_hidden(1)[:1].clone()      # the agent's initial state, moved to the device outside the capture

        def body():
            obs = [g.fresh() for g in batch["obs"]]
            fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
            h = h_row.expand(a.n * a.B, -1).contiguous()
            for t in range(a.T):
                _, h = learner.act(obs[t].fresh(), h, 0.05)
            return learner.update(fb)

        cyc = GraphedCycle(learner, body)
        for _ in range(max(1, a.warmup)):
            cyc()
        barrier()
        t_g = time.perf_counter()
        for _ in range(a.steps):
            out_g = cyc()
        barrier()
        el_g = time.perf_counter() - t_g
        graphed_leg = {"value": a.B * a.T * a.steps / el_g, "unit": "env-steps/s", "steps": a.steps, "ms_per_step": 1e3 * el_g / a.steps,
                       "loss": float(out_g["LossQ"]),
                       "what": "the same cycle captured ONCE into a hipGraph (graphs.GraphedCycle) and replayed: no launch gaps.  The capture "
                               "holds the stored batch's topology (edge counts, derived indices); a loop fed fresh observations replays a "
                               "cycle only over capacity-sized static graphs (graph.from_padded_obs(static=True), as GraphedAct / "
                               "GraphedUpdate do) - hence a separate figure, never `value`"}
    if rank == 0:
        env_steps = world * a.B * a.T * a.steps
        # SURVEY 8d names: C3 = BASELINE configs[2] (8 x 80, B = 4096); anything else is labelled by its own sizes
        label = {(8, 80, 4096): "C3", (4, 40, 1024): "C2", (16, 200, 1024): "C5 (per-GPU shard)"}.get((a.n, a.M, a.B), "custom")
        res = {
            "metric": metric_name(a), "value": env_steps / elapsed, "unit": "env-steps/s",
            "replay_ratio": 1,                 # the headline cycle trains each stored sequence once; the reference's own operating point
                                               # (run.py:55-57,:97: rho = 32) is `value_rho32` below, from the `replay_ratio_leg`
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}: {a.n} UBS x {a.M} GT, exp3 MADRQN (GATv2 obs-encoder + TarMAC), "
                                   f"B={a.B} env graphs/GPU, T={a.T}, D-{a.dist}, replay ratio 1: one step = {a.T} "
                                   f"act forwards + 1 update ({2 * a.T + 1} forwards + BPTT backward + AdamW)",
                       "global_batch": world * a.B, "seq_len": a.T, "parallelism": f"dp{world}",
                       **({} if backend == "nccl" else {"collective_backend": backend + " (test hook, not RCCL)"})},
            "loss": loss,
            "tuned_gemms": bool(tuned),        # recorded vendor-GEMM solutions (uav_bs_ctrl_amd/tuned/gemm_gfx950.csv) accepted by this box's hipBLASLt
            "shader_clock": clock_info,        # sustained clock of the timed region (the part runs at its power limit: DESIGN.md section 6)
            "rccl_ranks": (dist.get_world_size() if use_dist else 1),
            "replicas_identical": replicas_identical,   # parameter checksums (sum, sum of squares) min == max over the ranks after the timed steps
            "self_launched": os.environ.get("UAVGNN_BENCH_SELF_LAUNCHED") == "1",   # plain `python3 bench.py --gpus N` started its own ranks
            "rccl_transport": rccl_transport_summary(rccl_log) if rccl_log else None,   # only from a caller-provided NCCL_DEBUG=INFO log
            "peer_access": peer_access_summary(world) if use_dist else None,
            "collective_backend": (None if not use_dist else
                                   {"backend": dist.get_backend(), "rccl_version": ".".join(map(str, th.cuda.nccl.version())),
                                    "NCCL_DEBUG": os.environ.get("NCCL_DEBUG"),
                                    "note": "one process per GPU; set NCCL_DEBUG=INFO to see the transport (xGMI / P2P) RCCL picked"}),
            "collective_ms": (None if not coll_ms else {"per_update_mean": sum(coll_ms) / len(coll_ms), "max": max(coll_ms),
                                                        "calls": len(coll_ms), "bytes": 4 * learner.grads.flat.numel(),
                                                        "what": "all-reduce (sum) of the flat fp32 gradient buffer + the 1/G scaling, "
                                                                "HIP events on the compute stream"}),
            "params_checksum": [float(ck[0]), float(ck[1])],   # sum / sum of squares of the policy parameters after the timed steps
            "step_ms_device": [round(marks[i - 1].elapsed_time(marks[i]), 1) for i in range(1, len(marks))],
            # diagnostics of the timed region: cumulative host time at which each step was enqueued (a launch thread that falls behind the
            # device shows here, not in step_ms_device) and the device allocations (hipMalloc) the caching allocator made inside it
            "step_enqueued_at_ms": [round(1e3 * t, 1) for t in enq], "device_allocs_in_timed_region": int(alloc1 - alloc0),
            "device_allocs_by_step": [int(b - a_) for a_, b in zip([alloc0] + allocs_by_step[:-1], allocs_by_step)][:8],
        }
        # ---- roofline of the dominant message-passing kernel: K1 forward, BOTH relations (one fused launch) ----------
        k = ktimes.get("gatv2_hetero_fwd") or kfull.get("gatv2_hetero_fwd")   # graphed cycle: from the eager instrumented cycle
        if k:
            res["roofline"] = k1_roofline(k, a, a.dist, clock_mhz)
        res["graphed_cycle"] = False           # the timed steps are eager launches; the replayed cycle is `graphed_cycle_leg`
        if graphed_leg is not None:
            res["graphed_cycle_leg"] = graphed_leg
            res["value_graphed"] = graphed_leg["value"]
        # ---- the GEMM-shaped kernels by time share (the GRU cell is the largest kernel of a dense cycle): MFMA roofs ----------
        sec = []
        kc = kfull.get("gru_cell_fwd")
        if kc and kc["work"]:
            fl = sum(2.0 * N * 3 * H * (K_in + H) for (N, K_in, H, _) in kc["work"])       # fp32-equivalent FLOP
            kinds = {w[3] for w in kc["work"]}
            x3, h2 = kinds == {"bf16x3"}, kinds == {"f16x2"}
            tf = fl / (kc["total_ms"] * 1e-3) / 1e12
            mult = 6.0 if x3 else 3.0 if h2 else 1.0          # 16-bit MFMA products per fp32 product
            sec.append({"kernel": ("gru_cell_fwd_h2_kernel + split_planes_h2_kernel (K4, whole GRU cell; f16x2: exactly scaled two-term f16 splits, "
                                   "row maxima from the message kernel)" if h2 else
                                   "gru_cell_fwd_x3w8_kernel + split_planes_kernel (K4, whole GRU cell)" if x3 else
                                   "gru_cell_fwd_kernel (K4, whole GRU cell, fp32 MFMA)" if kinds == {"f32"} else
                                   "GRU cell, mixed kernels: " + ", ".join(sorted(kinds))),
                        "bound": "mfma", "launches": kc["count"], "avg_launch_ms": kc["avg_ms"],
                        "fp32_equivalent_tflops": tf,
                        "achieved": mult * tf, "peak": BF16_PEAK_TFLOPS if mult > 1 else FP32_PEAK_TFLOPS,
                        "unit": ("TFLOP/s (f16 MFMA: three products per fp32 product)" if h2 else
                                 "TFLOP/s (bf16 MFMA: six products per fp32 product)" if x3 else "TFLOP/s"),
                        "frac": (mult * tf / BF16_PEAK_TFLOPS) if mult > 1 else tf / FP32_PEAK_TFLOPS,
                        "share_of_step": kc["total_ms"] / (1e3 * instr_s)})
        for span, label, mult, what in (("gemm_x3", "gemm_nt_x3w8_kernel + split_matrix_kernel (dense layers, bf16x3: the f_aggr forward)", 6.0, "bf16"),
                                        ("gemm_h2", "gemm_nt_h2w8_kernel + split_h2_kernel (input-gradient products of the recurrent step and of f_aggr, "
                                                    "f16x2: row maxima from the producing kernels)", 3.0, "f16"),
                                        ("gemm_tn_h2", "gemm_tn_h2_kernel (weight gradients dW_ih, dW_hh, dW_aggr over the time-batched rows, f16x2 through "
                                                       "LDS transposing reads; rounds 2-5: the vendor's fp32 split-K GEMM)", 3.0, "f16")):
            kg = kfull.get(span)
            if kg and kg["work"]:
                fl = sum(2.0 * M * N * K for (M, N, K) in kg["work"])
                tf = fl / (kg["total_ms"] * 1e-3) / 1e12
                sec.append({"kernel": label, "bound": "mfma", "launches": kg["count"], "avg_launch_ms": kg["avg_ms"],
                            "fp32_equivalent_tflops": tf, "achieved": mult * tf, "peak": BF16_PEAK_TFLOPS,
                            "unit": f"TFLOP/s ({what} MFMA: {int(mult)} products per fp32 product)", "frac": mult * tf / BF16_PEAK_TFLOPS,
                            "share_of_step": kg["total_ms"] / (1e3 * instr_s)})
        km = kfull.get("tarmac_msg_fwd")
        if km and km["work"]:
            # K3a + K3b in one launch (csrc/tarmac_msg.hip): x and h read once, c written (+ proj and the x half of [x || c] on
            # training forwards); algorithmic bytes / launch time against the HBM peak
            by = sum(Nn * (2 * Hh * 4 + Mm * 4) + tr * Nn * (Mm + 2 * Kk) * 4 + xc * Nn * Hh * 4 for (Nn, Hh, Mm, Kk, tr, xc) in km["work"])
            g = by / (km["total_ms"] * 1e-3) / 1e9
            sec.append({"kernel": "tarmac_msg_fwd_k2_kernel (K3a + K3b, two wavefronts per row tile: projection GEMM on the bf16 matrix cores + per-tile attention, one launch)",
                        "bound": "hbm", "launches": km["count"], "avg_launch_ms": km["avg_ms"], "achieved": g, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": g / HBM_PEAK_GBS, "alg_bytes_per_launch": by / km["count"],
                        "share_of_step": km["total_ms"] / (1e3 * instr_s),
                        "note": "replaces two vendor GEMMs + talk_attn_env_fwd (52-56 us per C3 step); its phases - prologue, activation "
                                "stream, MFMAs, weight-slice traffic, attention tail - run one after the other in every workgroup "
                                "(tools/msg_probe.py ablations), so it sits at ~0.25 of the HBM peak, not at the ~13 us its traffic allows"})
        if sec:
            cal = mfma_calibration(device) if world == 1 else None
            if cal:
                # what the vendor's own bf16 GEMM sustains on THIS box in THIS run: the part runs MFMA-dense kernels at its power limit
                # (DESIGN.md section 5 / 6), so the nominal 2.5 PFLOP/s is not reachable in steady state by any kernel
                res["mfma_calibration"] = cal
                for e in sec:
                    if e.get("bound") == "mfma" and e["peak"] == BF16_PEAK_TFLOPS:
                        e["frac_of_vendor_bf16_gemm_rate"] = e["achieved"] / cal["tflops"]
            res["roofline_secondary"] = sec
        res["kernel_ms_per_launch"] = {n: round(v["avg_ms"], 4) for n, v in kfull.items()}
        res["instrumented_cycle"] = {"ms": 1e3 * instr_s,
                                     "note": "one extra cycle AFTER the timed steps with HIP events around every C-ABI launch: source of "
                                             "`roofline_secondary` and `kernel_ms_per_launch`.  The timed steps carry events around the "
                                             "graded kernel (`roofline`) only - timing every launch makes the rollout host-bound"}
        res["arithmetic"] = ("fp32 in / out / accumulate everywhere.  K3b, K5 and all pointwise kernels: fp32 FMA.  GEMM-shaped products run on the "
                             "16-bit matrix cores at fp32 accuracy, in two exact-split schemes.  f16x2 (round 6; csrc/gru_h2.hip, gemm_h2.hip, "
                             "gemm_tn_h2.hip): every row (column, for the weight gradients) of an operand is scaled by an exact power of two into "
                             "f16's range and split into hi + lo f16 terms (hi + lo = v to <= 2^-23 |v|); an fp32 product is the fp32-accumulated "
                             "sum of 3 exact f16 x f16 MFMA products (dropped term <= 2^-22 |a b|) - the GRU cell of the TarMAC step, the "
                             "input-gradient products of the recurrent step and of f_aggr, the f_aggr forward behind a K1 launch that leaves row maxima "
                             "(time-batched launches, rollout launches on dense degrees), the weight gradients dW_ih / dW_hh / dW_aggr / dWp; the "
                             "scales come from row maxima that the producing kernels leave on their way.  bf16x3 (rounds 2-5; csrc/gemm_x3.hip, "
                             "gatv2_hetero.hip, tarmac_msg.hip, gru_x3.hip): operands split EXACTLY into 3 bf16 terms, 6 exact bf16 x bf16 "
                             "products per fp32 product (dropped terms <= 2^-23 |a b|) - the f_aggr forward of rollout launches on env-realistic degrees, K1's score GEMM, the message "
                             "projection, GRU cells without a row-maxima producer.  Measured error vs fp64 of both schemes is at or BELOW the "
                             "vendor fp32 GEMM's on the same data (profiles/r06_h2_probe.txt, r06_h2_error_tables.txt, r06_gemm_tn_h2_probe.txt; "
                             "r02_gemm_x3_probe.txt, r03_gru_probe.txt).  A/B switches: UAVGNN_GRU_H2=0 / UAVGNN_GEMM_H2=0 / "
                             "UAVGNN_GEMM_TN_H2=0 fall back to bf16x3 / the vendor GEMM.  `fp32_mfma_leg` is the same cycle with ALL of them "
                             "switched to fp32 MFMA / vendor fp32 GEMMs (UAVGNN_GRU_X3=0 UAVGNN_GEMM_X3=0 UAVGNN_K1_BF16Z=0).")
        if world == 1 and not a.no_fp32_leg:
            saved_flags = (ops.GRU_X3, ops.GEMM_X3, ops.K1_BF16Z)     # a run started with UAVGNN_*=0 keeps its own setting afterwards
            ops.GRU_X3 = ops.GEMM_X3 = ops.K1_BF16Z = False
            try:
                eager_step()
                th.cuda.synchronize()
                gc.collect()
                gc.disable()
                t1 = time.perf_counter()
                for _ in range(5):
                    eager_step()
                th.cuda.synchronize()
                e1 = time.perf_counter() - t1
                gc.enable()
                res["fp32_mfma_leg"] = {"value": world * a.B * a.T * 5 / e1, "unit": "env-steps/s", "steps": 5,
                                        "ms_per_step": 1e3 * e1 / 5,
                                        "config": "UAVGNN_GRU_X3=0 UAVGNN_GEMM_X3=0 UAVGNN_K1_BF16Z=0: GRU cell on fp32 MFMA "
                                                  "(csrc/gru_fused.hip), every dense layer on the vendor fp32 GEMM, K1's score GEMM "
                                                  "on fp32 MFMA (csrc/gatv2_hetero_f32.hip): no bf16 instruction anywhere"}
            finally:
                ops.GRU_X3, ops.GEMM_X3, ops.K1_BF16Z = saved_flags
        if world == 1 and a.dist == "dense" and not a.no_env_leg:
            # ---- the HBM-bound regime of the graded kernel inside the default run: the same cycle on D-env degrees (94 % of the
            # agents see no GT - a random-policy rollout), a few steps, K1 forward timed live the same way
            env_batch = make_sequence(a.B, a.n, a.M, a.T, "env", device, seed=4321, distinct=a.distinct)

            def step_env():
                obs = [g.fresh() for g in env_batch["obs"]]
                fb = dict(env_batch, obs=obs, obs_all=env_batch["obs_all"].fresh(), obs_all_next=env_batch["obs_all_next"].fresh())
                h = learner.init_hidden(a.B)
                for t in range(a.T):
                    _, h = learner.act(obs[t].fresh(), h, 0.05)
                return learner.update(fb)
            ops.KERNEL_TIMER.reset(enabled=True, only=GRADED)
            step_env()
            th.cuda.synchronize()
            gc.collect()
            gc.disable()
            ops.KERNEL_TIMER.reset(enabled=True, only=GRADED)
            clock_e = ClockSampler(local)
            with clock_e:
                t1 = time.perf_counter()
                for _ in range(a.env_steps):
                    step_env()
                th.cuda.synchronize()
                e1 = time.perf_counter() - t1
            gc.enable()
            ke = ops.KERNEL_TIMER.summary().get("gatv2_hetero_fwd")
            ops.KERNEL_TIMER.enabled = False
            ce = clock_e.summary()
            if ke:
                r_env = k1_roofline(ke, a, "env", ce["mean_mhz"] if ce else None)
                r_env.update({"steps": a.env_steps, "ms_per_step": 1e3 * e1 / a.env_steps,
                              "env_steps_per_s": a.B * a.T * a.env_steps / e1, "shader_clock": ce,
                              "workload": f"the same cycle on D-env degrees (B={a.B}, {a.n} x {a.M}): {a.env_steps} timed steps after "
                                          "one warm-up, K1 forward between HIP events as in the timed region"})
                try:
                    r_env["standalone"] = k1_env_standalone(learner, env_batch["obs"][0].fresh(), a)
                    rp = k1_env_rocprof(r_env["standalone"].get("alg_bytes_per_launch", 70315280))
                    if rp is not None:
                        r_env["rocprof"], r_env["rocprof_frac"] = rp, rp["frac"]
                except Exception as e:   # noqa: BLE001 - a diagnostic leg never breaks the bench line
                    r_env["standalone"] = {"error": repr(e)}
                res["roofline_env"] = r_env
            del env_batch
        if world == 1 and not a.no_rho_leg:
            # the reference's own replay ratio (run.py:55-57,:97: 32 stored sequences per T steps of one environment):
            # T act forwards on B environments + ONE optimizer step over rho chunks of B sequences (gradient accumulation)
            rho = a.rho

            def step_rho(chunks):
                obs = [g.fresh() for g in batch["obs"]]
                fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
                h = learner.init_hidden(a.B)
                for t in range(a.T):
                    _, h = learner.act(obs[t].fresh(), h, 0.05)
                return learner.update([fb] * chunks)
            th.cuda.synchronize()
            gc.collect()
            th.cuda.empty_cache()   # like every leg after the headline: a clean caching allocator (the D-env leg's blocks and the pool of
            gc.disable()            # its captured graph cost this leg 30 % otherwise: 3.48 vs 2.65 s, profiles/r05_final_bench_dense.json)
            # ... and, like the headline, ONE untimed warm-up - a cycle with a single chunk: the chunks of an update run one after the other
            # through the same buffers, so this hands the empty allocator every block size of the timed cycle (timed cold, the leg measured
            # the box's hipMalloc: 2.2 s on one box, 3.1 s on another, same build - profiles/r06_final_bench_dense_run2.json)
            step_rho(1)
            th.cuda.synchronize()
            t1 = time.perf_counter()
            step_rho(rho)
            th.cuda.synchronize()
            e1 = time.perf_counter() - t1
            gc.enable()
            res["value_rho32" if rho == 32 else f"value_rho{rho}"] = a.B * a.T / e1    # env-steps/s at the reference's replay ratio
            res["replay_ratio_leg"] = {"rho": rho, "value": a.B * a.T / e1, "unit": "env-steps/s", "steps": 1,
                                       "ms_per_step": 1e3 * e1, "sequences_per_update": rho * a.B,
                                       "transitions_trained_per_s": rho * a.B * a.T / e1,
                                       "config": f"{a.T} act forwards on B environments + one optimizer step over {rho} chunks "
                                                 f"of B sequences (the same synthetic chunk {rho} times; gradients accumulated "
                                                 "in the flat buffer, ONE clip + AdamW + polyak)"}
        if world == 1 and not a.no_end_to_end:
            res["end_to_end"] = end_to_end(learner, a, device, "random")
            res["end_to_end_hotspot"] = end_to_end(learner, a, device, "hotspot")
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.n, a.M, a.dist)
        emit(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
