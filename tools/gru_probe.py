#!/usr/bin/env python
"""A/B of K4 at C3 size (N_a = 32768, GRUCell(320 -> 256)): the fused cell kernel (csrc/gru_fused.hip) vs the vendor
GEMMs + gate kernel path, forward without and with the saves for backward.  GPU box."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import enable_tuned_gemms, ops  # noqa: E402

print("recorded vendor-GEMM solutions:", enable_tuned_gemms())
dev = th.device("cuda")
N, K, H = 32768, 320, 256
cell = th.nn.GRUCell(K, H).to(dev)
inp, h = th.randn(N, K, device=dev), th.randn(N, H, device=dev)


def time_us(fn, reps=30):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 2.0 * N * 3 * H * (K + H)
ref = None
with th.no_grad():   # fp64 reference of h' on the same inputs: the error each variant makes
    c64 = th.nn.GRUCell(K, H).to(dev).double()
    c64.load_state_dict({k: v.double() for k, v in cell.state_dict().items()})
    ref = c64(inp[:4096].double(), h[:4096].double())
for name, fused, x3 in (("fused K4 kernel, bf16x3 MFMA", True, True), ("fused K4 kernel, fp32 MFMA", True, False),
                        ("vendor GEMMs + gate kernel", False, False)):
    ops.GRU_FUSED, ops.GRU_X3 = fused, x3
    with th.no_grad():
        t_inf = time_us(lambda: ops.gru_cell(inp, h, cell))
        err = (ops.gru_cell(inp[:4096], h[:4096], cell).double() - ref).abs()
    i_g = inp.clone().requires_grad_(True)
    t_tr = time_us(lambda: ops.gru_cell(i_g, h, cell))
    out = ops.gru_cell(i_g, h, cell)
    d = th.randn_like(out)
    t_bw = time_us(lambda: th.autograd.grad(out, [i_g] + list(cell.parameters()), d, retain_graph=True), reps=10)
    print(f"{name:30s}: forward no-grad {t_inf:7.1f} us = {fl / t_inf * 1e-6:6.1f} TFLOP/s | forward with saves {t_tr:7.1f} us | "
          f"backward {t_bw:7.1f} us | h' vs fp64: max {err.max().item():.2e} mean {err.mean().item():.2e}")
ops.GRU_FUSED, ops.GRU_X3 = True, True
