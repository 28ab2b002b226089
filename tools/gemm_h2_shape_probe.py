#!/usr/bin/env python
"""f16x2 NT GEMM (csrc/gemm_h2.hip) over a grid of shapes: what costs the time-batched launches their rate - the short contraction (a
tile's fixed costs against 8 or 16 slices of MFMAs) or the operand coming from HBM instead of the L2 / MALL?  GPU box.
    python tools/gemm_h2_shape_probe.py"""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import ops  # noqa: E402

dev = th.device("cuda")


def time_us(fn, reps):
    for _ in range(2):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("# y[M, N] = x[M, K] W[N, K]^T on the f16x2 kernel: us per call, fp32-equivalent TFLOP/s, GB/s of x + y")
for M in (32768, 262144, 1671168):
    for K, N in ((256, 512), (512, 256), (768, 256), (256, 256), (1024, 256)):
        x = th.randn(M, K, device=dev)
        W = th.randn(N, K, device=dev) / K ** 0.5
        rm = x.abs().amax(1)
        y = th.empty(M, N, device=dev)
        with ops.frozen_weights():
            f = lambda: ops.gemm_h2(x, W, rm, out=y)  # noqa: E731
            t = min(time_us(f, 20 if M < 10 ** 6 else 5) for _ in range(2))
        print(f"M {M:8d}  K {K:5d}  N {N:4d}: {t:9.1f} us  {2.0 * M * K * N / t * 1e-6:6.1f} TF  {(M * (K + N) * 4) / t * 1e-3:7.0f} GB/s   tiles {((M + 255) // 256) * ((N + 127) // 128)}")
        del x, y
