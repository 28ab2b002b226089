#!/usr/bin/env python
"""Weight-gradient GEMM dW = dy^T x at the layer shapes of the path: csrc/gemm_tn_x3.hip (bf16x3) vs the vendor's batched
split-K fp32 GEMM (what round 2 used), time per call in accumulate mode and error against float64.  GPU box."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import enable_tuned_gemms, ops  # noqa: E402


def time_us(fn, reps=20):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    enable_tuned_gemms()
    gen = th.Generator(device="cuda").manual_seed(0)
    print(f"{'shape (n, out, in)':28s} {'tn_x3 us':>9s} {'TF-eq':>7s} {'vendor us':>10s} {'TF':>7s} {'err x3':>9s} {'err vendor':>10s}")
    for n, Mo, Ko in ((32768, 768, 320), (32768, 768, 256), (32768, 96, 256), (32768, 9, 256), (32768, 256, 512),
                      (51 * 32768, 256, 512)):
        dy = th.randn(n, Mo, device="cuda", generator=gen) * 0.3
        x = th.randn(n, Ko, device="cuda", generator=gen)
        S = ops.WeightGradSink._chunks(n)
        pv = th.zeros(S, Mo, Ko, device="cuda")
        px = ops.gemm_tn_x3(dy, x)
        t_x = time_us(lambda: ops.gemm_tn_x3(dy, x, px, accumulate=True))
        t_v = time_us(lambda: pv.baddbmm_(dy.view(S, n // S, -1).transpose(1, 2), x.view(S, n // S, -1)))
        fl = 2.0 * n * Mo * Ko
        if n <= 32768:
            ref = dy.double().t() @ x.double()
            scale = dy.double().abs().t() @ x.double().abs()
            e_x = float(((ops.gemm_tn_x3(dy, x).sum(0).double() - ref).abs() / scale).max())
            e_v = float(((th.bmm(dy.view(S, n // S, -1).transpose(1, 2), x.view(S, n // S, -1)).sum(0).double() - ref).abs() / scale).max())
        else:
            e_x = e_v = float("nan")
        print(f"{str((n, Mo, Ko)):28s} {t_x:9.1f} {fl / t_x / 1e6:7.1f} {t_v:10.1f} {fl / t_v / 1e6:7.1f} {e_x:9.2e} {e_v:10.2e}  "
              f"(S = {px.shape[0]} / {S})")


if __name__ == "__main__":
    main()
