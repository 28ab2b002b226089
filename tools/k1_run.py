#!/usr/bin/env python
"""Launches ONLY the K1 forward kernel (both relations, fused launch) on the C3 workload a few times - the target of the
rocprofv3 --pmc passes (tools/pmc.sh).

    python tools/k1_run.py [--dist dense|env] [--reps 3] [--save] [--phases 3] [--no-image]
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
ap.add_argument("--phases", type=int, default=3)
ap.add_argument("--no-image", action="store_true", help="in-kernel prologue instead of the prepared parameter image (what a call outside a frozen_weights scope runs)")
a = ap.parse_args()
dev = th.device("cuda")
gen = th.Generator(device=dev)
gen.manual_seed(0)
th.manual_seed(0)
hb = synth_batch_gpu(a.B, 8, 80, a.dist, dev, gen)
xs, so = hb.relation_segments("seen")
xn, no = hb.relation_segments("near")
order = hb.relation_order("seen")
x_a = hb.agent_feat()
N = x_a.shape[0]
ps = []
for FS in (4, 2):
    c = GATv2Conv((FS, 2), 64, 4).to(dev)
    ps.append([t.detach().contiguous() for t in (c.fc_src.weight, c.fc_src.bias, c.fc_dst.weight, c.fc_dst.bias, c.attn,
                                                 c.res_fc.weight, c.res_fc.bias)])
out = th.empty(N, 512, device=dev)
a_s, a_n = th.empty(max(xs.shape[0], 1), 4, device=dev), th.empty(max(xn.shape[0], 1), 4, device=dev)
lib, st = L.lib(), L.stream()
image = th.empty(lib.uavgnn_gatv2_hetero_image_bytes(), dtype=th.uint8, device=dev)
assert lib.uavgnn_gatv2_hetero_prepare(L.ptr_array(ps[0]), L.ptr_array(ps[1]), 4, 64, 0.2, image.data_ptr(), st) == 0
head = (xs.data_ptr(), xs.shape[0], so.data_ptr(), L.ptr(order), xn.data_ptr(), xn.shape[0], no.data_ptr(), x_a.data_ptr(), N,
        L.ptr_array(ps[0]), L.ptr_array(ps[1]), 4, 64, 0.2)
tail = (out.data_ptr(), 512, a_s.data_ptr() if a.save else None, a_n.data_ptr() if a.save else None, a.phases, st)
for _ in range(a.reps):
    rc = lib.uavgnn_gatv2_hetero_fwd_phases(*head, *tail) if a.no_image else lib.uavgnn_gatv2_hetero_fwd_image(*head, image.data_ptr(), *tail)
    assert rc == 0
th.cuda.synchronize()
print("ok", N, xs.shape[0], xn.shape[0])
