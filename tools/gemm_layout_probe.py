#!/usr/bin/env python
"""Forward GEMMs of the recurrent step: y = x W^T with W stored [out, in] (TN for the BLAS) vs a pre-transposed copy
[in, out] (NN), both with TunableOp tuning on, operands rotated through a 512 MB buffer."""
import os

os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME="/tmp/probe.csv",
                  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="40", PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE="512")
import torch as th  # noqa: E402


def t(fn, n=30):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


N = 32768
for (K, O) in ((320, 768), (256, 768), (512, 256), (256, 96)):
    xs = [th.randn(N, K, device="cuda") for _ in range(8)]     # rotate inputs: nothing L2-resident
    W = th.randn(O, K, device="cuda")
    Wt = W.t().contiguous()
    b = th.randn(O, device="cuda")
    i = [0]

    def tn():
        i[0] = (i[0] + 1) % 8
        return th.addmm(b, xs[i[0]], W.t())

    def nn():
        i[0] = (i[0] + 1) % 8
        return th.addmm(b, xs[i[0]], Wt)
    fl = 2 * N * K * O
    a_, b_ = t(tn), t(nn)
    print(f"[{N} x {K}] x [{K} x {O}]:  W[out,in] (TN) {a_:7.1f} us = {fl / a_ / 1e6:6.1f} TF   W^T copy (NN) {b_:7.1f} us = {fl / b_ / 1e6:6.1f} TF")
