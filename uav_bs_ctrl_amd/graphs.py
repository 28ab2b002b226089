"""hipGraph capture of the rollout step and of the whole update.

At the reference's own operating point - ONE environment per ``act`` and 32 sequences of T = 50 steps per ``update``
(algos/madrqn/learner.py:69-80,:94-173; run.py:55-57) - the path is launch-bound: an ``act`` is ~20 kernels of a few
microseconds each and an ``update`` ~3000.  Every kernel of the path takes an explicit stream, allocates nothing and
keeps no state (include/uavgnn.h), the device-side graph builder has a ``static`` mode without a host round trip, the
exploration rate / learning rate / Adam step count live in device memory, and the update's tail is one launch
(uav_bs_ctrl_amd/optim.py) - so both calls capture into ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) and replay from
fixed-address input buffers.  Same arithmetic as the eager calls, kernel for kernel.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch as th

from . import _lib as L
from .graph import from_padded_obs


def _capture(graph, **kw):
    """``torch.cuda.graph`` for this module's captures.  With a process group alive the capture runs in THREAD-LOCAL error mode:
    ProcessGroupNCCL's watchdog thread polls the events of the collectives issued so far (``hipEventQuery``), and under the default
    global mode a query that lands inside another thread's capture window is an error - "operation not permitted when stream is
    capturing" - that terminates the process (seen once in ~8 runs of the world-size-1 RCCL test of ``GraphedUpdate``, whose warm-up
    updates leave all-reduces for the watchdog to retire).  Thread-local mode restricts the check to the capturing thread, which
    issues nothing but this library's launches."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        kw.setdefault("capture_error_mode", "thread_local")
    return th.cuda.graph(graph, **kw)


class _RngSnapshot:
    """The random state a warm-up advances besides the parameters: the learner's epsilon generator and the device {seed, step} pairs
    of DiscreteComm's in-kernel noise (``rng_state``).  Restored after the capture, so a graphed run starts from the state an eager
    run starts from.  (A generator registered with a capturing graph keeps its captured offset bookkeeping; only its state before
    the first replay is put back.)"""

    def __init__(self, learner):
        self.learner = learner
        self.gen = learner._gen.get_state()
        self.comm = [(m, m.rng_state.clone()) for net in (learner.policy_net, learner.target_net) for m in net.modules()
                     if isinstance(getattr(m, "rng_state", None), th.Tensor)]

    def restore(self):
        self.learner._gen.set_state(self.gen)
        for m, st in self.comm:
            m.rng_state.copy_(st)


class _PaddedObs:
    """Fixed-address padded observation buffers (the simulator's format, mubs_cov.py:215-242) of ``lead`` env steps."""

    def __init__(self, lead, n, M, device, with_comm=True):
        f = dict(dtype=th.float32, device=device)
        self.gt = th.zeros(*lead, n, M, 5, **f)
        self.ubs = th.zeros(*lead, n, max(n - 1, 0), 3, **f)
        self.agent = th.zeros(*lead, n, 2, **f)
        self.d_u2u = th.zeros(*lead, n, n, **f) if with_comm else None

    def load(self, gt, ubs, agent, d_u2u=None):
        self.gt.copy_(gt, non_blocking=True)
        self.ubs.copy_(ubs, non_blocking=True)
        self.agent.copy_(agent, non_blocking=True)
        if self.d_u2u is not None:
            self.d_u2u.copy_(d_u2u, non_blocking=True)


class GraphedAct:
    """``learner.act`` on B environments as one graph replay: device-side graph construction from padded observations,
    no-grad policy forward, epsilon-greedy selection (one draw per team, learner.py:75-78).

        ga = GraphedAct(learner, B, n, M, r_comm)
        acts, h = ga(gt, ubs, agent, d_u2u, h, eps)        # acts [B*n] int64, h' [B*n, H]; both are the graph's buffers
    """

    def __init__(self, learner, B: int, n: int, M: int, r_comm: float = float("inf"), warmup: int = 2):
        self.learner, self.B, self.n, self.M, self.r_comm = learner, B, n, M, r_comm
        dev = learner.device
        with_comm = learner.args.c is not None
        self.obs = _PaddedObs((B,), n, M, dev, with_comm)
        self.h_in = th.zeros(B * n, learner.args.hidden_size, dtype=th.float32, device=dev)
        self.eps = th.zeros(1, dtype=th.float32, device=dev)
        self._eps_host = None
        self.graph = th.cuda.CUDAGraph()
        if hasattr(self.graph, "register_generator_state"):
            self.graph.register_generator_state(learner._gen)
        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):              # warm-up on a side stream (allocator pools, lazy kernels) before capture
            for _ in range(warmup):
                self._body()
        th.cuda.current_stream().wait_stream(side)
        with _capture(self.graph):
            self.acts, self.h_out = self._body()

    @th.no_grad()
    def _body(self):
        lr = self.learner
        g = from_padded_obs(self.obs.gt, self.obs.ubs, self.obs.agent, self.obs.d_u2u, self.r_comm, static=True)
        logits, h = lr.policy_net(g, self.h_in)
        N = logits.shape[0]
        u = th.rand(self.B + N, device=lr.device, generator=lr._gen)
        acts = th.empty(N, dtype=th.int64, device=lr.device)
        logits = logits if logits.stride(1) == 1 else logits.contiguous()
        L.check(L.lib().uavgnn_eps_greedy_dev(logits.data_ptr(), logits.stride(0), N, lr.n_actions, lr.n_agents,
                                              u.data_ptr(), u.data_ptr() + 4 * self.B, self.eps.data_ptr(),
                                              acts.data_ptr(), L.stream()), "uavgnn_eps_greedy_dev")
        return acts, h

    def __call__(self, gt, ubs, agent, d_u2u, h, eps_thres: float):
        """Copies the observation into the graph's buffers and replays.  A producer that writes ``self.obs.gt / .ubs /
        .agent / .d_u2u`` and ``self.h_in`` in place (e.g. a device-side simulator) passes gt=None and skips the copies."""
        if gt is not None:
            self.obs.load(gt, ubs, agent, d_u2u)
        if h is not None and h.data_ptr() != self.h_in.data_ptr():
            self.h_in.copy_(h if h.shape[0] == self.h_in.shape[0] else h.expand_as(self.h_in), non_blocking=True)
        if eps_thres != self._eps_host:
            self.eps.copy_(th.tensor([eps_thres], dtype=th.float32), non_blocking=True)
            self._eps_host = eps_thres
        self.graph.replay()
        return self.acts, self.h_out


class GraphedUpdate:
    """``learner.update`` on B stored sequences of T transitions as one graph replay: graphs of all T+1 steps rebuilt on
    the device from the padded observations of the sampled batch (``SequenceReplay.mem`` layout), time-batched encoder,
    2T+1 forwards, BPTT backward, clip + AdamW + polyak.

        gu = GraphedUpdate(learner, B, T, n, M, r_comm)
        out = gu(batch)      # batch: gt [B,T+1,n,M,5], ubs, agent, d_u2u, h [B,T+1,n,H], act [B,T,n], rew [B,T,rd], done [B,T,1]

    Data-parallel runs (``learner.needs_collective()``): the gradient all-reduce is NOT captured.  The update is cut at
    its only collective into TWO graphs - ``accumulate`` (graph construction, 2T+1 forwards, backward into the flat gradient
    buffer) and ``apply`` (clip + AdamW + polyak) - with the RCCL all-reduce of the flat buffer issued eagerly on the same
    stream between the two replays: the capture never depends on what the communicator does under stream capture, and a
    rank that replays while another is still capturing cannot dead-lock inside a captured collective.
    """

    def __init__(self, learner, B: int, T: int, n: int, M: int, r_comm: float = float("inf"), rew_dim: Optional[int] = None,
                 warmup: int = 2):
        assert learner.fused_tail, "graph capture needs the device-resident update tail (CUDA learner)"
        self.learner, self.B, self.T, self.n, self.M, self.r_comm = learner, B, T, n, M, r_comm
        dev, H = learner.device, learner.args.hidden_size
        rd = n if rew_dim is None else rew_dim
        self.obs = _PaddedObs((T + 1, B), n, M, dev, True)        # time-major: step t of every sequence is contiguous
        self.h0 = th.zeros(B * n, H, dtype=th.float32, device=dev)
        self.h1 = th.zeros(B * n, H, dtype=th.float32, device=dev)
        self.acts = th.zeros(T, B * n, 1, dtype=th.int64, device=dev)
        self.rews = th.zeros(T, B, rd, dtype=th.float32, device=dev)
        self.dones = th.zeros(T, B, 1, dtype=th.float32, device=dev)
        self.graph = th.cuda.CUDAGraph()
        # warm-up updates run for real (they would move the parameters): snapshot and restore around them
        learner.optimizer.sync_lr()      # a learning rate the scheduler moved since the last sync is part of the snapshot, not undone by it
        snap = [t.clone() for t in (learner.flat.flat, learner.flat_target, learner.optimizer.m, learner.optimizer.v,
                                    learner.optimizer.hyper)]
        rng = _RngSnapshot(learner)
        self.split = learner.needs_collective()
        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        th.cuda.current_stream().wait_stream(side)
        if self.split:
            th.cuda.synchronize()          # the warm-up updates' all-reduces are complete before a capture window opens
            self.graph_tail = th.cuda.CUDAGraph()
            with _capture(self.graph):
                self.out = self.learner.accumulate(self._batch())
            with _capture(self.graph_tail, pool=self.graph.pool()):
                self.learner.apply()
        else:
            with _capture(self.graph):
                self.out = self._body()
        th.cuda.synchronize()
        for dst, src in zip((learner.flat.flat, learner.flat_target, learner.optimizer.m, learner.optimizer.v,
                             learner.optimizer.hyper), snap):
            dst.copy_(src)
        rng.restore()
        learner.invalidate_weight_cache()

    def _body(self) -> Dict:
        return self.learner.update(self._batch())

    def _batch(self) -> Dict:
        T, B, n, M = self.T, self.B, self.n, self.M
        o = self.obs
        obs = [from_padded_obs(o.gt[t], o.ubs[t], o.agent[t], o.d_u2u[t], self.r_comm, static=True) for t in range(T + 1)]
        flat = lambda x, lo: x[lo:].reshape((-1,) + x.shape[2:])  # noqa: E731
        obs_all = from_padded_obs(flat(o.gt, 0), flat(o.ubs, 0), flat(o.agent, 0), None, self.r_comm, static=True)
        obs_next = from_padded_obs(flat(o.gt, 1), flat(o.ubs, 1), flat(o.agent, 1), None, self.r_comm, static=True)
        return dict(obs=obs, obs_all=obs_all, obs_all_next=obs_next, h0=self.h0, h1=self.h1, acts=self.acts,
                    rews=self.rews, dones=self.dones)

    def load(self, m: Dict[str, th.Tensor]) -> None:
        """m: a gathered batch in ``SequenceReplay.mem`` layout (leading dims [B, T+1] / [B, T])."""
        B, T, n = self.B, self.T, self.n
        self.obs.load(m["gt"].transpose(0, 1), m["ubs"].transpose(0, 1), m["agent"].transpose(0, 1),
                      m["d_u2u"].transpose(0, 1))
        self.h0.copy_(m["h"][:, 0].reshape(B * n, -1), non_blocking=True)
        self.h1.copy_(m["h"][:, 1].reshape(B * n, -1), non_blocking=True)
        self.acts.copy_(m["act"].permute(1, 0, 2).reshape(T, B * n, 1), non_blocking=True)
        self.rews.copy_(m["rew"].permute(1, 0, 2), non_blocking=True)
        self.dones.copy_(m["done"].permute(1, 0, 2), non_blocking=True)

    def __call__(self, m: Optional[Dict[str, th.Tensor]] = None) -> Dict:
        if m is not None:
            self.load(m)
        self.learner.optimizer.sync_lr()
        self.graph.replay()
        if self.split:
            self.learner.grads.all_reduce_mean_(self.learner.group)
            self.graph_tail.replay()
        self.learner.invalidate_weight_cache()   # the replay moved the parameters without passing through learner.apply()
        return self.out


class GraphedCycle:
    """A whole acting / training cycle on FIXED-ADDRESS inputs - e.g. T ``learner.act`` calls on stored graphs followed by one
    ``learner.update`` - as one graph replay.  For batches of a few thousand agents (BASELINE config 2: 4 x 40, B = 1024) the
    ~4000 launches of a cycle are 5-20 us each and the launch thread, not the device, sets the pace; the replay removes the gaps.

        cyc = GraphedCycle(learner, body)      # body(): learner calls only, every input at a fixed device address, no host
        out = cyc()                            # round trip; returns body()'s value (the graph's own buffers)

    ``body`` runs ``warmup`` times for real before the capture (allocator pools, lazy kernels, plane caches); parameters,
    target and optimiser state are restored afterwards.  Single-process only: a data-parallel update holds a
    collective (``GraphedUpdate`` cuts the capture there).  The learning rate is pushed to the device before each replay
    and the rollout's weight-plane store is emptied after it (the replay moved the parameters without passing through
    ``learner.apply``)."""

    def __init__(self, learner, body, warmup: int = 2):
        assert learner.fused_tail, "graph capture needs the device-resident update tail (CUDA learner)"
        assert not learner.needs_collective(), "a data-parallel update cannot be captured whole: use GraphedUpdate"
        self.learner, self.body = learner, body
        self.graph = th.cuda.CUDAGraph()
        if hasattr(self.graph, "register_generator_state"):
            self.graph.register_generator_state(learner._gen)
        # the warm-up cycles run for real (an update inside `body` moves the parameters): snapshot and restore around them, so a
        # graphed run starts from the state an eager run starts from
        learner.optimizer.sync_lr()      # a learning rate the scheduler moved since the last sync is part of the snapshot, not undone by it
        state = (learner.flat.flat, learner.flat_target, learner.optimizer.m, learner.optimizer.v, learner.optimizer.hyper)
        snap = [t.clone() for t in state]
        rng = _RngSnapshot(learner)      # the warm-up's epsilon draws and DiscreteComm noise steps are undone as well
        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):
            for _ in range(warmup):
                body()
        th.cuda.current_stream().wait_stream(side)
        learner.invalidate_weight_cache()
        with _capture(self.graph):
            self.out = body()
        th.cuda.synchronize()
        for dst, src in zip(state, snap):
            dst.copy_(src)
        rng.restore()
        learner.invalidate_weight_cache()

    def __call__(self):
        self.learner.optimizer.sync_lr()
        self.graph.replay()
        self.learner.invalidate_weight_cache()
        return self.out
