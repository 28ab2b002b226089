#!/usr/bin/env python
"""Launches ONLY the K1 kernels on the C3 workload a few times (target of rocprofv3 --pmc passes).

    python tools/k1_run.py [--dist dense|env] [--reps 3] [--save] [--bwd]
"""
import argparse
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch_gpu  # noqa: E402
from uav_bs_ctrl_amd import _lib as L  # noqa: E402
from uav_bs_ctrl_amd.agents.gnn_agents import GATv2Conv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="dense")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--B", type=int, default=4096)
ap.add_argument("--save", action="store_true")
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--rel", default="seen")
a = ap.parse_args()
dev = th.device("cuda")
gen = th.Generator(device=dev)
gen.manual_seed(0)
th.manual_seed(0)
hb = synth_batch_gpu(a.B, 8, 80, a.dist, dev, gen)
FS = 4 if a.rel == "seen" else 2
x_src, off = hb.relation_segments(a.rel)
_o = hb.relation_order(a.rel)
ORDER = _o.data_ptr() if _o is not None else None
x_a = hb.agent_feat()
N, E = x_a.shape[0], x_src.shape[0]
conv = GATv2Conv((FS, 2), 64, 4).to(dev)
p = [t.detach().contiguous() for t in (conv.fc_src.weight, conv.fc_src.bias, conv.fc_dst.weight, conv.fc_dst.bias,
                                       conv.attn, conv.res_fc.weight, conv.res_fc.bias)]
out = th.empty(N, 256, device=dev)
a_save = th.empty(max(E, 1), 4, device=dev)
lib, st = L.lib(), L.stream()
d_out = th.randn(N, 256, device=dev)
g = [th.empty_like(t) for t in p]
wsb = lib.uavgnn_gatv2_bwd_workspace_bytes(FS, 256)
ws = th.empty(wsb // 4, device=dev)
for _ in range(a.reps):
    rc = lib.uavgnn_gatv2_fwd(x_src.data_ptr(), x_src.shape[0], FS, x_a.data_ptr(), 2, off.data_ptr(), ORDER, N, *[t.data_ptr() for t in p], 4, 64,
                              0.2, out.data_ptr(), 256, a_save.data_ptr() if (a.save or a.bwd) else None, st)
    assert rc == 0
    if a.bwd:
        rc = lib.uavgnn_gatv2_bwd(x_src.data_ptr(), x_src.shape[0], FS, x_a.data_ptr(), 2, off.data_ptr(), ORDER, N,
                                  *[t.data_ptr() for t in p[:5]], 4, 64, 0.2, out.data_ptr(), d_out.data_ptr(), 256,
                                  a_save.data_ptr(), *[t.data_ptr() for t in g], ws.data_ptr(), wsb, st)
        assert rc == 0
th.cuda.synchronize()
print("ok", N, E)
