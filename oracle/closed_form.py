"""Closed-form tensors for fixtures: no RNG state, no weight files.  TEST INFRASTRUCTURE ONLY.

``W[i, j] = amp * sin(0.37 i + 0.11 j + phi)`` on the tensor flattened to 2-D (leading dim x rest); ``phi`` differs
per tensor so that no two parameters coincide (SURVEY 8c "wiring goldens").
"""
import math

import torch as th


def closed_form_tensor(shape, phi: float, amp: float = 0.1, dtype=None) -> th.Tensor:
    dtype = dtype or th.get_default_dtype()
    shape = tuple(shape)
    rows = shape[0] if len(shape) > 0 else 1
    cols = 1
    for s in shape[1:]:
        cols *= s
    i = th.arange(rows, dtype=th.float64).view(-1, 1)
    j = th.arange(cols, dtype=th.float64).view(1, -1)
    return (amp * th.sin(0.37 * i + 0.11 * j + phi)).view(shape).to(dtype)


def fill_closed_form(module, amp_weight: float = 0.25, amp_bias: float = 0.1) -> None:
    """Overwrites every parameter of ``module`` in ``named_parameters()`` order; phi = 1 + index * pi/7."""
    with th.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            amp = amp_bias if p.dim() == 1 else amp_weight
            p.copy_(closed_form_tensor(p.shape, 1.0 + k * math.pi / 7, amp, p.dtype))
