#!/usr/bin/env python
"""K1 forward (`seen`, C3 dense) with the SAME input every call vs inputs rotated through 16 buffers (672 MB > the
256 MB Infinity Cache), and right after a burst of large fp32 GEMMs (clock / power state of the bench)."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch_gpu  # noqa: E402
from uav_bs_ctrl_amd import _lib as L  # noqa: E402
from uav_bs_ctrl_amd.agents.gnn_agents import GATv2Conv  # noqa: E402

dev = th.device("cuda")
gen = th.Generator(device=dev).manual_seed(0)
g = synth_batch_gpu(4096, 8, 80, "dense", dev, gen)
x_src, off = g.relation_segments("seen")
x_a, N = g.agent_feat(), g.num_nodes("agent")
xs = [x_src.clone() + 0.001 * i for i in range(16)]
outs = [th.empty(N, 512, device=dev) for _ in range(16)]
th.manual_seed(0)
conv = GATv2Conv((4, 2), 64, 4).cuda()
p = [t.detach().contiguous() for t in (conv.fc_src.weight, conv.fc_src.bias, conv.fc_dst.weight, conv.fc_dst.bias,
                                       conv.attn, conv.res_fc.weight, conv.res_fc.bias)]
st = L.stream()


ORDER = [None]


def k1(i):
    rc = L.lib().uavgnn_gatv2_fwd(xs[i].data_ptr(), x_src.shape[0], 4, x_a.data_ptr(), 2, off.data_ptr(), ORDER[0], N,
                                  *[t.data_ptr() for t in p], 4, 64, 0.2, outs[i].data_ptr(), 512, None, st)
    assert rc == 0


def timed(fn, n):
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    th.cuda.synchronize()
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for _ in range(5):
    k1(0)
print(f"same input        {timed(lambda i: k1(0), 64):7.1f} us")
print(f"rotating inputs   {timed(lambda i: k1(i % 16), 64):7.1f} us")
order = g.fresh().relation_order("seen")
ORDER[0] = order.data_ptr()
print(f"with hand-out order (identity here: all degrees equal)  {timed(lambda i: k1(i % 16), 64):7.1f} us")
perm = th.randperm(N, device=dev).to(th.int32)
ORDER[0] = perm.data_ptr()
print(f"with a random permutation as order                      {timed(lambda i: k1(i % 16), 64):7.1f} us")
ORDER[0] = None
A = th.randn(32768, 320, device=dev)
B = th.randn(768, 320, device=dev)
ev = []
for rep in range(20):
    for _ in range(6):
        th.mm(A, B.t())
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    k1(rep % 16)
    b.record()
    ev.append((a, b))
th.cuda.synchronize()
t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(f"after 6 GEMMs     median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}")


def bracket(pre, n=40):
    ev = []
    for i in range(n):
        pre(i)
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        k1(i % 16)
        b.record()
        ev.append((a, b))
    th.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]


tiny = th.zeros(4, device=dev)
big = th.empty(64 << 20, device=dev)      # 256 MB
print("event-bracketed K1 after: nothing          median %7.1f  min %7.1f" % bracket(lambda i: None))
print("event-bracketed K1 after: a 16-byte fill    median %7.1f  min %7.1f" % bracket(lambda i: tiny.fill_(1.0)))
print("event-bracketed K1 after: a 256 MB fill     median %7.1f  min %7.1f" % bracket(lambda i: big.fill_(1.0)))
print("event-bracketed K1 after: one GEMM          median %7.1f  min %7.1f" % bracket(lambda i: th.mm(A, B.t())))
