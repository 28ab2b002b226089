"""Probe: column sums (bias gradients) of [N, C] fp32 at the bench's shapes - torch.sum vs staged variants."""
import torch as th

def t(fn, n=30):
    for _ in range(5): fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

N = 32768
for C in (5, 96, 256, 768):
    x = th.randn(N, C, device="cuda")
    acc = th.zeros(C, device="cuda")
    r = {"sum0+add": t(lambda: acc.add_(x.sum(0)))}
    for S in (16, 64, 256, 1024):
        accS = th.zeros(S, C, device="cuda")
        r[f"view{S}.sum1+add"] = t(lambda: accS.add_(x.view(S, N // S, C).sum(1)))
    if C == 768:
        xs = x[:, 512:]
        acc2 = th.zeros(256, device="cuda")
        r["strided n-part sum0"] = t(lambda: acc2.add_(xs.sum(0)))
        for S in (64, 256):
            accS = th.zeros(S, 256, device="cuda")
            r[f"strided view{S}"] = t(lambda: accS.add_(xs.reshape(S, N // S, 256).sum(1)) if False else accS.add_(x.view(S, N // S, C)[:, :, 512:].sum(1)))
    ones = th.ones(N, device="cuda")
    r["mv"] = t(lambda: th.mv(x.t(), ones))
    print(C, {k: round(v, 1) for k, v in r.items()}, flush=True)
