#!/usr/bin/env python
"""bf16x3 GEMM kernel (csrc/gemm_x3.hip) vs the vendor fp32 GEMM PyTorch calls, at the GEMM shapes of a C3 cycle, with the
error of both against an fp64 product of the same operands.  GPU box."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import enable_tuned_gemms, ops  # noqa: E402

print("recorded vendor-GEMM solutions:", enable_tuned_gemms())
dev = th.device("cuda")
th.manual_seed(0)


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
    return e0.elapsed_time(e1) / reps * 1e3


N = 32768
cases = [("f_aggr forward (rollout)      x[N,512] W[256,512]^T", N, 256, 512, False),
         ("projections forward          x[N,256] W[96,256]^T ", N, 96, 256, False),
         ("GRU d_inp = d_gi W_ih        dy[N,768] W[768,320] ", N, 320, 768, True),
         ("GRU dh += d_gh W_hh          dy[N,768] W[768,256] ", N, 256, 768, True),
         ("f_aggr d_x (time-batched/51) dy[N,256] W[256,512] ", N, 512, 256, True),
         ("f_aggr forward, T+1 = 51 steps", 51 * N, 256, 512, False)]
# the accumulating launch of the GRU backward (Y += X W): the epilogue reads Y
_a = th.randn(N, 768, device=dev)
_W = th.randn(768, 256, device=dev) * 0.06
_y = th.randn(N, 256, device=dev)
for flags, what in ((8, "4-wave"), (0, "8-wave")):
    ops.GEMM_X3_FLAGS = flags
    print(f"GRU dh += d_gh W_hh WITH the accumulate epilogue ({what}): {time_us(lambda: ops.gemm_x3(_a, _W, True, out=_y, accumulate=True)):7.1f} us"
          f" | without: {time_us(lambda: ops.gemm_x3(_a, _W, True, out=_y)):7.1f} us")
ops.GEMM_X3_FLAGS = 0
for name, M, n_out, K, tr in cases:
    a = th.randn(M, K, device=dev)
    W = (th.randn(K, n_out, device=dev) if tr else th.randn(n_out, K, device=dev)) * 0.06
    f_x3 = lambda: ops.gemm_x3(a, W, tr)
    f_v = (lambda: th.mm(a, W)) if tr else (lambda: th.mm(a, W.t()))
    from uav_bs_ctrl_amd import _lib as L
    ops.GEMM_X3_FLAGS = 8   # UAVGNN_GEMM_TILE_128
    t_x3_w4 = time_us(f_x3)
    ops.GEMM_X3_FLAGS = 4     # eight waves, staging interleaved with the MFMAs
    t_x3_il = time_us(f_x3)
    y_il = f_x3()
    ops.GEMM_X3_FLAGS = 0     # the default: eight waves, the staging of a slice as a block in front of its MFMAs
    t_x3, t_v = time_us(f_x3), time_us(f_v)
    same = bool(th.equal(y_il, f_x3()))
    rows = slice(0, 2048)
    ref = a[rows].double() @ (W.double() if tr else W.double().t())
    scale = a[rows].double().abs() @ (W.double().abs() if tr else W.double().abs().t())
    e_x3 = ((f_x3()[rows].double() - ref).abs() / scale)
    e_v = ((f_v()[rows].double() - ref).abs() / scale)
    fl = 2.0 * M * n_out * K
    print(f"{name}: [4-wave 128x128 tiles {t_x3_w4:8.1f} us] [8-wave, interleaved staging {t_x3_il:8.1f} us, bit-identical {same}] bf16x3 {t_x3:8.1f} us = {fl / t_x3 * 1e-6:6.1f} TFLOP/s | vendor {t_v:8.1f} us = {fl / t_v * 1e-6:6.1f} TFLOP/s | "
          f"x{t_v / t_x3:.2f} | error / sum|a b|: bf16x3 max {e_x3.max().item():.1e} mean {e_x3.mean().item():.1e}, "
          f"vendor max {e_v.max().item():.1e} mean {e_v.mean().item():.1e}")
