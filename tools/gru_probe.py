#!/usr/bin/env python
"""A/B of K4 at C3 size (N_a = 32768, GRUCell(320 -> 256)), four arms, each timed in TWO passes over all arms (the first pass
absorbs one-time costs: library heuristics, allocator growth, clocks; the second is the one to read):

  fused x3      one-kernel cell on the bf16 matrix cores (csrc/gru_x3.hip)                       GRU_FUSED, GRU_X3, GEMM_X3
  fused fp32    one-kernel cell on fp32 MFMA (csrc/gru_fused.hip)                                 GRU_FUSED, GEMM_X3
  unfused x3    two bf16x3 GEMMs (csrc/gemm_x3.hip, 768 columns tile by 128) + gate kernel         GEMM_X3
  vendor        two vendor fp32 GEMMs (recorded solutions) + gate kernel                          (nothing)

Per arm: forward without saves (no-grad call class), forward with saves, backward (d inp, d h, all parameters), and the
training total forward-with-saves + backward.  The two fused arms share _GruCellFused.backward; the two unfused arms share
autograd's linear / gate-kernel backward - differences inside a pair are measurement noise.  GPU box."""
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
with th.no_grad():   # fp64 reference of h' on the same inputs: the error each variant makes
    c64 = th.nn.GRUCell(K, H).to(dev).double()
    c64.load_state_dict({k: v.double() for k, v in cell.state_dict().items()})
    ref = c64(inp[:4096].double(), h[:4096].double())
ARMS = (("fused x3", True, True, True), ("fused fp32", True, False, True), ("unfused x3", False, False, True),
        ("vendor", False, False, False))
for rnd in (1, 2):
    print(f"--- pass {rnd}")
    for name, fused, x3, gx3 in ARMS:
        ops.GRU_FUSED, ops.GRU_X3, ops.GEMM_X3 = fused, x3, gx3
        with th.no_grad():
            t_inf = time_us(lambda: ops.gru_cell(inp, h, cell))
            err = (ops.gru_cell(inp[:4096], h[:4096], cell).double() - ref).abs()
        i_g, h_g = inp.clone().requires_grad_(True), h.clone().requires_grad_(True)
        t_tr = time_us(lambda: ops.gru_cell(i_g, h_g, cell))
        out = ops.gru_cell(i_g, h_g, cell)
        d = th.randn_like(out)
        t_bw = time_us(lambda: th.autograd.grad(out, [i_g, h_g] + list(cell.parameters()), d, retain_graph=True), reps=10)

        def both():
            o = ops.gru_cell(i_g, h_g, cell)
            th.autograd.grad(o, [i_g, h_g] + list(cell.parameters()), d)
        t_both = time_us(both, reps=10)
        print(f"{name:11s}: fwd no-grad {t_inf:6.1f} us ({fl / t_inf * 1e-6:5.1f} TF) | fwd + saves {t_tr:6.1f} | bwd {t_bw:6.1f} | "
              f"fwd + saves + bwd {t_both:6.1f} | h' vs fp64 max {err.max().item():.2e} mean {err.mean().item():.2e}")
ops.GRU_FUSED, ops.GRU_X3, ops.GEMM_X3 = True, True, True
