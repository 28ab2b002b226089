"""Flat parameter / gradient / optimizer-state buffers and the one-launch update tail.

The reference's update ends with ``clip_grad_value_(policy_net.parameters(), 1)``, ``AdamW.step()`` and a Python loop
over parameter pairs for the polyak target update (algos/madrqn/learner.py:157-166) - on a GPU ~15 multi-tensor
launches over 40 small tensors.  Here every parameter of the Q-network (and mixer) is a VIEW into one contiguous fp32
buffer, likewise the gradients, both Adam moments and the target network, so the whole tail is ONE HIP launch
(csrc/optim.hip ``uavgnn_adamw_polyak``) and the data-parallel exchange is ONE all-reduce of the flat gradient buffer.

``FusedAdamW`` is a ``torch.optim.Optimizer`` (so ``LambdaLR`` drives it) whose ``state_dict`` has the layout of
``torch.optim.AdamW``'s: a checkpoint written by either side loads into the other (learner.py:175-201).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch as th

from . import _lib as L

ALIGN = 4   # floats: every parameter starts on a 16-byte boundary (the kernels fetch parameters with 16-byte loads)


def _groups_of(modules: Sequence[th.nn.Module]):
    """Parameter groups that want to be contiguous in memory (e.g. TarMAC's stacked projection weight)."""
    out = []
    for mod in modules:
        for sub in mod.modules():
            fn = getattr(sub, "_projection_params", None)
            if fn is not None:
                out.extend([list(g) for g in fn()])
    return out


class FlatParams:
    """All parameters of ``modules`` as views of ONE flat buffer (``.flat``), in ``parameters()`` order except that the
    members of a contiguity group follow the group's first member back to back.  ``mirror(other_modules)`` lays a second
    set of modules (the target networks) out identically."""

    def __init__(self, modules: Sequence[th.nn.Module]):
        self.modules = list(modules)
        params = [p for m in self.modules for p in m.parameters()]
        groups = _groups_of(self.modules)
        gid = {id(p): gi for gi, g in enumerate(groups) for p in g
               if all(q.numel() % ALIGN == 0 for q in g)}
        placed, self.offsets, o = set(), {}, 0
        for p in params:
            if id(p) in placed:
                continue
            members = groups[gid[id(p)]] if id(p) in gid else [p]
            for q in members:
                self.offsets[id(q)] = o
                placed.add(id(q))
                o += q.numel()
            o = (o + ALIGN - 1) // ALIGN * ALIGN
        self.params, self.numel = params, o
        self.index_of = [self.offsets[id(p)] for p in params]         # by position in parameters() order
        dev = params[0].device
        self.flat = th.zeros(o, dtype=th.float32, device=dev)
        self._adopt(params, self.flat)

    def _adopt(self, params, flat):
        with th.no_grad():
            for p, off in zip(params, self.index_of):
                view = flat[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view

    def mirror(self, modules: Sequence[th.nn.Module]) -> th.Tensor:
        """Same layout for another set of modules with identical parameter shapes (target nets); returns its flat buffer."""
        params = [p for m in modules for p in m.parameters()]
        assert [tuple(p.shape) for p in params] == [tuple(p.shape) for p in self.params]
        flat = th.zeros_like(self.flat)
        self._adopt(params, flat)
        return flat

    def views(self, flat: th.Tensor) -> List[th.Tensor]:
        return [flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.index_of)]

    def span(self, n_modules: int) -> int:
        """Flat length covered by the first n_modules modules (they are laid out first)."""
        k = sum(1 for m in self.modules[:n_modules] for _ in m.parameters())
        if k == len(self.params):
            return self.numel
        return min(self.index_of[k:])

    def intact(self) -> bool:
        """Every parameter still IS its slice of the flat buffer (``module.to()`` or ``p.data = ...`` would break that)."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.index_of))


class FusedAdamW(th.optim.Optimizer):
    """AdamW over a ``FlatParams`` with gradient value clipping of the first ``n_clip`` flat elements and the polyak update
    of a mirrored target buffer folded into the same launch."""

    def __init__(self, fp: FlatParams, grads_flat: th.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, clip: float = 0.0, n_clip: int = 0, target_flat: Optional[th.Tensor] = None,
                 polyak: float = 1.0):
        super().__init__(fp.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                         maximize=False, foreach=None, capturable=False, differentiable=False,
                                         fused=None, decoupled_weight_decay=True))
        self.fp, self.g, self.target = fp, grads_flat, target_flat
        self.clip, self.n_clip, self.polyak = float(clip), int(n_clip), float(polyak)
        self.m, self.v = th.zeros_like(fp.flat), th.zeros_like(fp.flat)
        self.hyper = th.tensor([lr, 0.0], dtype=th.float32, device=fp.flat.device)     # {lr, step count}
        self._lr_on_device = float(lr)
        self._steps = 0
        self._point_state()

    def _point_state(self):
        for p, ma, va in zip(self.fp.params, self.fp.views(self.m), self.fp.views(self.v)):
            self.state[p] = dict(step=th.tensor(float(self._steps)), exp_avg=ma, exp_avg_sq=va)

    def sync_lr(self):
        """Push a changed learning rate (LambdaLR edits param_groups on the host) to the device - OUTSIDE any capture."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.hyper[0:1].copy_(th.tensor([lr], dtype=th.float32), non_blocking=True)
            self._lr_on_device = lr

    @th.no_grad()
    def step(self, closure=None):
        if not th.cuda.is_current_stream_capturing():
            self.sync_lr()
        g0 = self.param_groups[0]
        self.hyper[1:2].add_(1.0)                # device-side step count: replays of a captured update keep counting
        self._steps += 1
        b1, b2 = g0["betas"]
        L.check(L.lib().uavgnn_adamw_polyak(self.fp.flat.data_ptr(), self.g.data_ptr(), self.m.data_ptr(),
                                            self.v.data_ptr(), L.ptr(self.target), self.fp.numel, self.n_clip,
                                            self.hyper.data_ptr(), float(b1), float(b2), float(g0["eps"]),
                                            float(g0["weight_decay"]), self.clip, self.polyak, L.stream()),
                "uavgnn_adamw_polyak")

    def state_dict(self):
        steps = float(self.hyper[1])             # authoritative (graph replays advance it without the host counter)
        self._steps = int(steps)
        for st in self.state.values():
            st["step"] = th.tensor(steps)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # fills self.state with fresh tensors: fold them back into the flat buffers
        steps = 0.0
        with th.no_grad():
            for p, ma, va in zip(self.fp.params, self.fp.views(self.m), self.fp.views(self.v)):
                st = self.state.get(p)
                if st:
                    ma.copy_(st["exp_avg"])
                    va.copy_(st["exp_avg_sq"])
                    steps = max(steps, float(st["step"]))
            self.hyper[1:2].fill_(steps)
        self._steps = int(steps)
        self._lr_on_device = float("nan")
        self.sync_lr()
        self._point_state()
