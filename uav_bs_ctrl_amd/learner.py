"""Batched, device-resident counterpart of the reference's ``MultiAgentQLearner`` (algos/madrqn/learner.py:14-201).

The reference's learner is a Python host harness around one env and a deque of DGLGraphs; it is not shipped.  This is
the build's own caller of the hot path, reproducing the same call pattern and numerics so that the fixtures captured
from the reference's ``update()`` pin it:

  * ``act``    - ``policy_net(obs, h)`` under ``no_grad`` -> argmax, epsilon-greedy with ONE draw per team
                 (learner.py:69-80), vectorised over B environments and without a device->host sync;
  * ``update`` - the 2T+1-forward BPTT pattern with stored-state initialisation (learner.py:110-128), gather of
                 Q(s,a), double-Q target, MSE over [T, B, n] (learner.py:134-154), ``clip_grad_value_(.., 1)``,
                 AdamW, polyak averaging of the target net (learner.py:157-166).

Data parallelism: one process per GPU, independent env batches per rank, replicated parameters; the ONLY
exchange is one all-reduce (RCCL over xGMI; ``nccl`` backend) of the Q-network gradients, kept in ONE flat buffer
(632 425 floats = 2.53 MB for exp3-TarMAC) so it is a single collective call, averaged BEFORE the element-wise clip
(so the result equals the single-GPU step on the concatenated batch).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch as th
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .agents import REGISTRY as agent_REGISTRY


class FlatGradBuffer:
    """All gradients of a module as views into one contiguous fp32 buffer (one collective instead of one per tensor;
    the reference's unused helper all-reduces per parameter: utils/mpi_pytorch.py:19-26).  ``offsets`` / ``numel``: the
    layout of an ``optim.FlatParams`` (gradient i sits where parameter i sits); default: densely packed."""

    def __init__(self, params: List[th.nn.Parameter], offsets: Optional[List[int]] = None, numel: Optional[int] = None):
        self.params = [p for p in params if p.requires_grad]
        if offsets is None:
            offsets, o = [], 0
            for p in self.params:
                offsets.append(o)
                o += p.numel()
            numel = o
        assert len(offsets) == len(self.params)
        self.offsets = list(offsets)
        dev = self.params[0].device
        self.flat = th.zeros(numel, dtype=th.float32, device=dev)
        self.force_collective = False   # run the collective even at world size 1 (RCCL smoke test / bench --force-dist)
        self.collective_events = None   # a list: all_reduce_mean_ appends a HIP event pair per call (bench.py `collective_ms`)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero_(self):
        self.flat.zero_()
        for p, o in zip(self.params, self.offsets):   # re-attach (zero_grad(set_to_none=True) would detach the views)
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                p.grad = self.flat[o:o + p.numel()].view_as(p)

    def all_reduce_mean_(self, group=None):
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.force_collective):
            timed = self.collective_events is not None and self.flat.is_cuda
            if timed:      # HIP events around the ONE collective of the data path (bench.py: `collective_ms`)
                e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if dist.get_world_size(group) > 1:
                self.flat.div_(dist.get_world_size(group))
            if timed:
                e1.record()
                self.collective_events.append((e0, e1))


def broadcast_parameters(module: th.nn.Module, src: int = 0, group=None, force: bool = False) -> None:
    """Start-up sync (counterpart of the reference's unused ``sync_params``, utils/mpi_pytorch.py:29-35): one broadcast
    of a flat copy of every parameter and buffer."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers())]
    flat = th.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    for t in tensors:
        t.copy_(flat[o:o + t.numel()].view_as(t))
        o += t.numel()


class MultiAgentQLearner:
    def __init__(self, env_info: dict, args, process_group=None):
        self.args = args
        self.device = th.device(args.device)
        self.obs_shape = env_info["obs_shape"]
        self.n_actions = env_info["n_actions"]
        self.n_agents = env_info["n_agents"]
        self.group = process_group

        self.policy_net = self._build_agent().to(self.device)
        self.target_net = self._build_agent().to(self.device)
        broadcast_parameters(self.policy_net, 0, process_group)
        self.target_net.load_state_dict(self.policy_net.state_dict())
        self.target_net.eval()
        for p in self.target_net.parameters():
            p.requires_grad_(False)
        self.params = list(self.policy_net.parameters())
        self.mixer = None
        if getattr(args, "mixer", False):   # QMIX (learner.py:35-40); every launcher of the reference sets mixer=False
            from copy import deepcopy

            from .agents.qmix import QMixer
            self.mixer = QMixer(env_info["state_shape"], self.n_agents, args).to(self.device)
            broadcast_parameters(self.mixer, 0, process_group)
            self.target_mixer = deepcopy(self.mixer)
            for p in self.target_mixer.parameters():
                p.requires_grad_(False)
            self.params += list(self.mixer.parameters())

        ep_limit = env_info.get("episode_limit")
        self.max_seq_len = args.max_seq_len if getattr(args, "max_seq_len", None) is not None else ep_limit
        self.gamma, self.polyak = args.gamma, args.polyak
        self.batch_size = getattr(args, "batch_size", None)
        self.double_q = args.double_q
        self.n_policy = sum(p.numel() for p in self.policy_net.parameters())
        self.fused_tail = self.device.type == "cuda"
        if self.fused_tail:
            # parameters, gradients, Adam moments and the target network as views of flat buffers: the update's tail
            # (clip + AdamW + polyak) is ONE launch, the DP exchange ONE all-reduce (uav_bs_ctrl_amd/optim.py)
            from .optim import FlatParams, FusedAdamW
            mods = [self.policy_net] + ([self.mixer] if self.mixer is not None else [])
            tmods = [self.target_net] + ([self.target_mixer] if self.mixer is not None else [])
            self.flat = FlatParams(mods)
            self.flat_target = self.flat.mirror(tmods)
            self.grads = FlatGradBuffer(self.flat.params, self.flat.index_of, self.flat.numel)
            self.optimizer = FusedAdamW(self.flat, self.grads.flat, lr=args.lr, clip=1.0, n_clip=self.flat.span(1),
                                        target_flat=self.flat_target, polyak=args.polyak)
        else:       # host-side harness (CPU tests of the data-parallel logic): stock optimizer, same arithmetic
            self.optimizer = th.optim.AdamW(self.params, lr=args.lr)
            self.grads = FlatGradBuffer(self.params)
        # lr annealing (learner.py:51-54; `anneal_lr` defaults to True in madrqn/config.py:36): the caller steps
        # ``learner.lr_scheduler`` once per epoch as run.py:107-108 does.  (The reference passes verbose=True, which
        # torch >= 2.7 rejects; the schedule itself is identical.)
        self.anneal_lr = bool(getattr(args, "anneal_lr", False))
        if self.anneal_lr:
            self.lr_scheduler = th.optim.lr_scheduler.LambdaLR(self.optimizer,
                                                               lr_lambda=lambda epoch: max(0.4, 1 - epoch / 100))
        self._rollout_planes = {}     # ops.frozen_weights store of `act`
        self._policy_params_for_fingerprint = tuple(self.policy_net.parameters())
        self._rollout_fingerprint = None
        self._gen = th.Generator(device=self.device)
        self._gen.manual_seed(int(getattr(args, "seed", 0)) + 7919 * (dist.get_rank() if dist.is_initialized() else 0))

    def _build_agent(self):
        return agent_REGISTRY["gnn"](self.obs_shape, self.n_actions, self.args)   # learner.py:62-67 ('gnn' arm)

    def init_hidden(self, batch_size: int = 1) -> th.Tensor:
        """[n_agents * batch_size, H] on the learner's device (learner.py:82-83).  The agent's one-row initial state (on
        the CPU, as the reference returns it) is moved to the device FIRST and expanded there: expanding on the host
        materialises B * n * H floats in pageable memory and copies them synchronously - 33 MB and ~10 ms of a blocked
        launch thread per rollout at C3 size (tools/prof_act_host.py)."""
        h0 = self.policy_net.init_hidden()
        if h0.device != self.device:
            h0 = h0.to(self.device, non_blocking=True)
        return h0.expand(self.n_agents * batch_size, -1).contiguous()

    # ---- rollout --------------------------------------------------------------------------------------------------
    @th.no_grad()
    def act(self, obs, h: th.Tensor, eps_thres: float):
        """obs: HeteroBatch of B envs (B*n agents).  Returns (acts [B*n] int64 on device, h').  One epsilon draw per
        team, as in learner.py:75-78."""
        obs, h = obs.to(self.device), h.to(self.device)
        # between two optimiser steps the policy's parameters stand still: weight planes / the K1 parameter image are built by
        # the first rollout step and reused by the others (cleared by apply / load_checkpoint / invalidate_weight_cache)
        # The store is keyed by device addresses (+ the version counter of the tensor a kernel is handed), but TarMAC's stacked
        # projection weight is a `.data` view with its OWN counter: an in-place write to f_val / f_sign / f_que (load_state_dict,
        # p.copy_()) bumps the parameters' counters and not the view's.  So the store is additionally guarded by the version
        # counters of ALL policy parameters (30 attribute reads per act): any torch-visible in-place write empties it.
        if self._rollout_planes is not None:
            fp = tuple(p._version for p in self._policy_params_for_fingerprint)
            if fp != self._rollout_fingerprint:
                self._rollout_planes.clear()
                self._rollout_fingerprint = fp
        with ops.frozen_weights(self._rollout_planes):
            logits, h = self.policy_net(obs, h)
        N = logits.shape[0]
        B = N // self.n_agents
        if logits.is_cuda:
            # one uniform per team + one per agent in ONE generator call, then argmax / compare / select in one HIP
            # launch (csrc/act_select.hip) instead of six tiny ones
            u = th.rand(B + N, device=self.device, generator=self._gen)
            acts = th.empty(N, dtype=th.int64, device=self.device)
            logits = logits if logits.stride(1) == 1 else logits.contiguous()
            L.check(L.lib().uavgnn_eps_greedy(logits.data_ptr(), logits.stride(0), N, self.n_actions, self.n_agents,
                                              u.data_ptr(), u.data_ptr() + 4 * B, float(eps_thres), acts.data_ptr(),
                                              L.stream()), "uavgnn_eps_greedy")
            return acts, h
        greedy = logits.argmax(1)
        explore = th.rand(B, device=self.device, generator=self._gen) <= eps_thres
        rand = th.randint(self.n_actions, greedy.shape, device=self.device, generator=self._gen)
        acts = th.where(explore.repeat_interleave(self.n_agents), rand, greedy)
        return acts, h

    # ---- replay -----------------------------------------------------------------------------------------------------
    def cache(self, buffer, obs: Dict, h, state, act, rew, next_obs: Dict, next_h, next_state, done, bad_mask,
              staged: bool = False) -> None:
        """learner.py:82-92 for E parallel environments on the device: one transition per environment is pushed into
        ``buffer`` (a ``replay.SequenceReplay``) with the reference's three rules -
          * ``share_reward``: every agent is credited the team MEAN of the step's rewards (``rew.mean()``), stored as one value;
          * the stored ``done`` is muted when the episode ended by its time limit: ``done = (1 - bad_mask) * done``;
          * the next hidden state is zeroed by the RAW done flag: ``next_h = (1 - done) * next_h`` (an episode that ends -
            for whatever reason - hands no recurrent state to the next one).
        obs / next_obs: the padded observation fields of the replay (gt, ubs, agent, d_u2u), each [E, ...]; h / next_h
        [E * n, H] or [E, n, H]; state / next_state [E, state_dim] or None; act [E * n] or [E, n]; rew [E, n] (or [E, 1] /
        [E]); done, bad_mask [E], [E, 1] or scalars.  staged: the observation half (obs fields, h, state) was written with
        ``buffer.stage_obs`` before the simulator overwrote its buffers - only act / rew / done / next_* are pushed."""
        E, n = buffer.n_envs, self.n_agents
        dev = buffer.device
        f32 = lambda x: th.as_tensor(x, dtype=th.float32, device=dev)   # noqa: E731
        done_raw = f32(done).reshape(-1, 1).expand(E, 1)
        bad = f32(0.0 if bad_mask is None else bad_mask).reshape(-1, 1).expand(E, 1)
        rew = f32(rew).reshape(E, -1)
        if getattr(self.args, "share_reward", False):
            rew = rew.mean(1, keepdim=True)
        tr = dict(act=th.as_tensor(act, device=dev).reshape(E, n).long(), rew=rew, done=(1.0 - bad) * done_raw,
                  next_h=(1.0 - done_raw).unsqueeze(2) * f32(next_h).reshape(E, n, -1))
        if not staged:
            tr["h"] = f32(h).reshape(E, n, -1)
        for k in ("gt", "ubs", "agent", "d_u2u"):
            if not staged:
                tr[k] = obs[k]
            tr["next_" + k] = next_obs[k]
        if state is not None:
            if not staged:
                tr["state"] = f32(state).reshape(E, -1)
            tr["next_state"] = f32(next_state).reshape(E, -1)
        buffer.push(tr)

    # ---- training -------------------------------------------------------------------------------------------------
    def loss(self, batch: Dict) -> tuple:
        """Forward part of ``update`` (learner.py:110-154).  batch: obs (list of T+1 HeteroBatch of B envs each),
        h0 / h1 [B*n, H] (stored hidden states of the first two steps), acts [T, B*n, 1] int64,
        rews [T, B, n or 1], dones [T, B, 1]; optional obs_all = the T+1 observation graphs batched in time-major
        order (``graph.batch(obs)``), which switches on the time-batched encoder."""
        obs = batch["obs"]
        T = len(obs) - 1
        h, h_targ = batch["h0"], batch["h1"]
        agent_out, target_out = [], []
        obs_all = batch.get("obs_all")
        if obs_all is not None and hasattr(self.policy_net, "encode"):
            # Time-batched encoder: the observation encoder does not depend on h, so all T+1 steps are encoded by ONE
            # call per network (one K1 launch per relation over (T+1) N_a destinations, one f_aggr GEMM); only the
            # recurrent part (comm block, GRU, head) walks the sequence.  Same arithmetic as 2T+1 separate forwards.
            N = h.shape[0]
            x_pol = self.policy_net.encode(obs_all)
            with th.no_grad():
                x_tgt = self.target_net.encode(batch.get("obs_all_next") or obs_all.slice_agents(N, (T + 1) * N))
            # per-step views + the slices of ONE gradient buffer the steps' backward passes write into (no stack)
            xs, slots = ops.time_split(x_pol, T + 1)
            xt = x_tgt.view(T, N, -1)
            if ops.GRAD_SINK is not None and th.is_grad_enabled() and ops.SEQ_STAGING:
                # the weight / bias gradients of the T + 1 recurrent steps are reduced ONCE, from time-batched buffers
                ops.GRAD_SINK.begin_sequence(T + 1, N, x_pol)
            for t in range(T):
                logits, h = self.policy_net.step(obs[t], xs[t], h, slots[t])
                agent_out.append(logits)
                with th.no_grad():
                    nxt, h_targ = self.target_net.step(obs[t + 1], xt[t], h_targ)
                    target_out.append(nxt)
            logits, h = self.policy_net.step(obs[T], xs[T], h, slots[T])
            agent_out.append(logits)
        else:
            for t in range(T):
                logits, h = self.policy_net(obs[t], h)
                agent_out.append(logits)
                with th.no_grad():
                    nxt, h_targ = self.target_net(obs[t + 1], h_targ)
                    target_out.append(nxt)
            logits, h = self.policy_net(obs[T], h)
            agent_out.append(logits)
        agent_out, target_out = th.stack(agent_out), th.stack(target_out)

        qvals = agent_out[:-1].gather(2, batch["acts"])
        if self.double_q:
            next_acts = agent_out[1:].detach().argmax(2, keepdim=True)
            next_vals = target_out.gather(2, next_acts)
        else:
            next_vals = target_out.max(2, keepdim=True)[0]
        B = batch["rews"].shape[1]
        qvals = qvals.view(T, B, self.n_agents)
        next_vals = next_vals.view(T, B, self.n_agents)
        if self.mixer is not None:                       # learner.py:145-148 (batch["states"]: [T+1, B, state_dim])
            qvals = self.mixer(qvals, batch["states"][:-1])
            next_vals = self.target_mixer(next_vals, batch["states"][1:])
        rews, dones = batch["rews"].expand_as(next_vals), batch["dones"].expand_as(next_vals)
        target = rews + self.gamma * (1 - dones) * next_vals
        return _mse(qvals, target), agent_out, target_out

    def update(self, batch) -> Dict:
        """One gradient step = ``accumulate`` (loss + backward into the flat gradient buffer) -> the ONE collective of the
        data path -> ``apply`` (clip + AdamW + polyak).  ``batch``: a batch dict, or a LIST of equally sized batch dicts
        whose gradients are accumulated before the one optimizer step (replay ratio rho = len(list): the reference
        consumes 32 stored sequences per T environment steps of ONE environment - run.py:55-57,:97 - i.e. rho = 32; at B
        environments per GPU that is rho chunks of B sequences).  The loss is the MSE over ALL chunks (the mean of the
        chunk means), so the step equals the step on the concatenated batch."""
        out = self.accumulate(batch)
        self.grads.all_reduce_mean_(self.group)           # the only collective of the data path
        self.apply()
        return out

    def needs_collective(self) -> bool:
        return bool(dist.is_available() and dist.is_initialized() and
                    (dist.get_world_size(self.group) > 1 or self.grads.force_collective))

    def accumulate(self, batch) -> Dict:
        """Zero the flat gradient buffer, then loss + backward of every chunk into it (learner.py:110-157)."""
        chunks = batch if isinstance(batch, (list, tuple)) else [batch]
        if len(chunks) == 0:
            raise ValueError("accumulate: empty list of batches")
        sizes = {tuple(b["acts"].shape) for b in chunks}
        if len(sizes) > 1:      # the mean of chunk means is the mean over all sequences only for equally sized chunks
            raise ValueError(f"accumulate: chunks of different sizes {sorted(sizes)}")
        self.grads.zero_()
        # weight gradients of the fused recurrent step are accumulated in place across the T+1 steps (and across the
        # chunks) and folded into the flat gradient buffer once (ops.WeightGradSink); everything else reaches it through
        # autograd
        ops.GRAD_SINK = sink = ops.WeightGradSink() if self.device.type == "cuda" else None
        loss = None
        try:
            with ops.frozen_weights():      # nothing changes a parameter until apply(): weight planes are split once
                for b in chunks:
                    loss_b, agent_out, _ = self.loss(b)
                    if len(chunks) > 1:
                        loss_b = loss_b / len(chunks)
                    loss_b.backward()
                    if sink is not None:
                        sink.end_sequence()
                    loss = loss_b.detach() if loss is None else loss + loss_b.detach()
            if sink is not None:
                sink.flush()
        finally:
            ops.GRAD_SINK = None
        return dict(LossQ=loss, QVals=agent_out.detach())      # QVals: the LAST chunk's (LossQ: the mean over all chunks)

    def invalidate_weight_cache(self) -> None:
        """Drops what ``act`` derived from the policy's parameters (bf16 weight planes, the K1 parameter image).  The learner
        calls it wherever IT changes them (apply, load_checkpoint; graphs.GraphedUpdate after a replay); code that writes
        ``policy_net`` parameters behind the learner's back (``p.data`` writes - they bump no version counter -, raw-pointer
        kernels) must call it too.  ``load_state_dict`` / in-place torch ops ON THE PARAMETERS are caught by ``act`` itself: it
        compares the version counters of every policy parameter with those the store was filled under (the cache keys alone
        would not do - TarMAC's stacked projection weight is a ``.data`` view whose own counter never moves)."""
        if self._rollout_planes is not None:     # None: no store across calls (every `act` builds its own)
            self._rollout_planes.clear()
        self._rollout_fingerprint = None

    def apply(self) -> None:
        """clip_grad_value_(policy_net.parameters(), 1) (the mixer is NOT clipped, learner.py:159) + AdamW step + polyak
        of target net / target mixer (learner.py:157-166)."""
        self.invalidate_weight_cache()
        if self.fused_tail:     # one launch over the flat buffers
            if not self.flat.intact():
                raise L.UavGnnError("a parameter was moved out of the learner's flat buffer (module.to() / p.data = ... "
                                    "after the learner was built): rebuild the learner")
            self.optimizer.step()
            return
        self.grads.flat[:self.n_policy].clamp_(-1.0, 1.0)
        self.optimizer.step()
        with th.no_grad():                                # polyak (learner.py:163-166)
            pt = list(self.target_net.parameters())
            th._foreach_mul_(pt, self.polyak)
            th._foreach_add_(pt, list(self.policy_net.parameters()), alpha=1 - self.polyak)
            if self.mixer is not None:
                mt = list(self.target_mixer.parameters())
                th._foreach_mul_(mt, self.polyak)
                th._foreach_add_(mt, list(self.mixer.parameters()), alpha=1 - self.polyak)

    # ---- checkpoints (same keys as learner.py:175-201) -------------------------------------------------------------
    def save_checkpoint(self, path: str, stamp: dict) -> None:
        ck = dict(stamp)
        ck["model_state_dict"] = self.policy_net.state_dict()
        ck["optimizer_state_dict"] = self.optimizer.state_dict()
        if self.mixer is not None:
            ck["mixer_state_dict"] = self.mixer.state_dict()
        if self.anneal_lr:
            ck["lr_scheduler_state_dict"] = self.lr_scheduler.state_dict()
        # the {seed, step} pair of DiscreteComm's in-kernel noise generator: an extra key, NOT a module buffer (the state_dict
        # names are the reference's contract), so that a resumed run continues its noise stream
        rng = {name: m.rng_state.cpu() for name, m in self.policy_net.named_modules() if getattr(m, "rng_state", None) is not None}
        if rng:
            ck["comm_rng_state"] = rng
        th.save(ck, path)

    def load_checkpoint(self, path: str) -> dict:
        ck = th.load(path, map_location=self.device)
        self.invalidate_weight_cache()
        self.policy_net.load_state_dict(ck["model_state_dict"])
        self.target_net.load_state_dict(self.policy_net.state_dict())
        if "optimizer_state_dict" in ck:
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if self.mixer is not None and "mixer_state_dict" in ck:
            self.mixer.load_state_dict(ck["mixer_state_dict"])
            self.target_mixer.load_state_dict(self.mixer.state_dict())
        if self.anneal_lr and "lr_scheduler_state_dict" in ck:
            self.lr_scheduler.load_state_dict(ck["lr_scheduler_state_dict"])
        for name, m in self.policy_net.named_modules():
            if name in ck.get("comm_rng_state", {}) and hasattr(m, "rng_state"):
                m.rng_state = ck["comm_rng_state"][name].to(self.device)
        return dict(epoch=ck.get("epoch"), t=ck.get("t"))


def _mse(x: th.Tensor, y: th.Tensor) -> th.Tensor:
    """F.mse_loss(x, y) (learner.py:154: the mean of squared differences) as a TWO-STAGE sum: row sums of a [S, numel / S] view, then
    the sum of the S row sums.  torch's full reduction of a large tensor to one scalar runs as several thread blocks that meet at a
    semaphore which a ``memset`` in front of the kernel clears; inside a replayed hipGraph of a whole cycle (graphs.GraphedCycle, ~4000
    nodes) that reduction's final write did not always land once the device had been synchronised between replays - the training step was
    right (parameters bit-identical to eager over 200 cycles), the reported LossQ was stale (found on the C2 bench line,
    tools/gc_loss_probe.py).  Neither stage here needs the semaphore path: one block per row, then one block.  On this build
    F.mse_loss also returns its scalar as a view of the [T, B, n] buffer of squared differences, which this avoids as well."""
    d = x - y
    n = d.numel()
    if n == 0:
        return d.sum()
    S = 1
    while S < 1024 and n % (2 * S) == 0 and n // (2 * S) >= 256:
        S *= 2
    return d.square().view(S, n // S).sum(1).sum() / n


def params_checksum(module: th.nn.Module) -> th.Tensor:
    """Cheap replica-consistency probe for DP runs: sum and sum of squares of all parameters (float64)."""
    flat = th.cat([p.detach().double().reshape(-1) for p in module.parameters()])
    return th.stack([flat.sum(), flat.square().sum()])
