#!/usr/bin/env python
"""K3b micro-benchmark at the bench shape (B=4096 graphs x 8 agents, complete talk, msg 64, key 16): per-graph vs
per-destination kernels, forward with / without the fused x copy, backward."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch_gpu  # noqa: E402
from uav_bs_ctrl_amd import _lib as L, ops  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


dev = th.device("cuda")
gen = th.Generator(device=dev).manual_seed(0)
B, n, M, K, H = 4096, 8, 64, 16, 256
g = synth_batch_gpu(B, n, 80, "dense", dev, gen)
N = B * n
proj = th.randn(N, M + 2 * K, device=dev)
x = th.randn(N, H, device=dev)
inp = th.empty(N, H + M, device=dev)
d_inp = th.randn(N, H + M, device=dev)
d_proj = th.empty(N, M + 2 * K, device=dev)
off, src = g.talk_csc()
a_save = th.empty(src.shape[0], device=dev)
ld = M + 2 * K
for name, env in (("per-graph", ops._talk_env(g, M, K)), ("per-destination", None)):
    tr = g.talk_transpose() if env is None else None
    for xc in (False, True):
        us = t(lambda: ops._launch_talk_fwd(env, proj.data_ptr() + 4 * M, ld, proj.data_ptr() + 4 * (M + K), ld,
                                            proj.data_ptr(), ld, K, M, off, src, N, 1.0 / K, inp.data_ptr() + 4 * H, H + M,
                                            a_save.data_ptr(), x.data_ptr() if xc else None, H, H if xc else 0))
        print(f"{name:16s} fwd x_copy={int(xc)}  {us:7.1f} us")
    us = t(lambda: ops._launch_talk_bwd(env, proj.data_ptr() + 4 * M, ld, proj.data_ptr() + 4 * (M + K), ld, proj.data_ptr(),
                                        ld, K, M, off, src, tr, N, 1.0 / K, a_save, d_inp.data_ptr() + 4 * H, H + M,
                                        d_proj.data_ptr() + 4 * M, ld, d_proj.data_ptr() + 4 * (M + K), ld,
                                        d_proj.data_ptr(), ld))
    print(f"{name:16s} bwd           {us:7.1f} us")
us = t(lambda: inp[:, :H].copy_(x))
print(f"torch strided copy of x into [x || c]: {us:7.1f} us")
