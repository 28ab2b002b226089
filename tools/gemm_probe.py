#!/usr/bin/env python
"""Times the vendor fp32 GEMM (what F.linear / torch.mm dispatch to) on the shapes of the exp3 agent at N_a = 32768."""
import torch as th


def t(fn, reps=20):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / reps


N = 32768
dev = "cuda"
print(f"{'op':38s} {'ms':>8s} {'TFLOP/s':>8s}")
for name, K, O in (("f_aggr 512->256", 512, 256), ("gi 320->768", 320, 768), ("gh 256->768", 256, 768),
                   ("proj-x 256->96", 256, 96), ("f_out 256->9", 256, 9)):
    x, W, b = th.randn(N, K, device=dev), th.randn(O, K, device=dev), th.randn(O, device=dev)
    dy = th.randn(N, O, device=dev)
    fl = 2 * N * K * O
    ms = t(lambda: th.addmm(b, x, W.t()))
    print(f"fwd  addmm  {name:26s} {ms:8.4f} {fl / ms / 1e9:8.1f}")
    ms = t(lambda: th.mm(dy, W))
    print(f"bwd  dX=dY W {name:25s} {ms:8.4f} {fl / ms / 1e9:8.1f}")
    ms = t(lambda: th.mm(dy.t(), x))
    print(f"bwd  dW plain {name:24s} {ms:8.4f} {fl / ms / 1e9:8.1f}")
    for S in (8, 16, 32, 64):
        ms = t(lambda: th.bmm(dy.view(S, N // S, O).transpose(1, 2), x.view(S, N // S, K)).sum(0))
        print(f"bwd  dW splitK S={S:<3d} {name:20s} {ms:8.4f} {fl / ms / 1e9:8.1f}")
