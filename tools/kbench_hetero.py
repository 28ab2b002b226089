#!/usr/bin/env python
"""K1 forward of the whole encoder: fused launch (gatv2_hetero.hip) vs the two per-relation launches.  GPU box.

    python tools/kbench_hetero.py [--B 4096] [--reps 30]
Rooflines: bytes = 16 E_seen + 8 E_near + N (8 + 8 + 2048) [+ 16 (E_seen + E_near) when saving attention],
flops = 3360 E_seen + 2320 E_near + 7168 N.
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


def time_ms(fn, reps):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--M", type=int, default=80)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dists", default="dense,env,zero")
    a = ap.parse_args()
    dev = th.device("cuda")
    gen = th.Generator(device=dev)
    gen.manual_seed(0)
    th.manual_seed(0)
    lib, st = L.lib(), L.stream()
    print(f"{'case':44s} {'us':>8s} {'alg GB/s':>9s} {'%HBM':>6s} {'TFLOP/s':>8s} {'%fp32':>6s}")
    for dist in a.dists.split(","):
        hb = synth_batch_gpu(a.B, a.n, a.M, dist, dev, gen)
        x_a = hb.agent_feat()
        N = x_a.shape[0]
        xs, so = hb.relation_segments("seen")
        xn, no = hb.relation_segments("near")
        order = hb.relation_order("seen")
        convs = [GATv2Conv((4, 2), 64, 4).to(dev), GATv2Conv((2, 2), 64, 4).to(dev)]
        ps = []
        for c in convs:
            with th.no_grad():
                for b in (c.fc_src.bias, c.fc_dst.bias, c.res_fc.bias):
                    b.normal_(0, 0.1)
            ps.append([t.detach().contiguous() for t in (c.fc_src.weight, c.fc_src.bias, c.fc_dst.weight, c.fc_dst.bias,
                                                         c.attn, c.res_fc.weight, c.res_fc.bias)])
        out = th.empty(N, 512, device=dev)
        out2 = th.empty(N, 512, device=dev)
        a_s, a_n = th.empty(max(xs.shape[0], 1), 4, device=dev), th.empty(max(xn.shape[0], 1), 4, device=dev)
        a_s2, a_n2 = th.empty_like(a_s), th.empty_like(a_n)

        def fused(o, sv_s, sv_n):
            rc = lib.uavgnn_gatv2_hetero_fwd(xs.data_ptr(), xs.shape[0], so.data_ptr(), L.ptr(order), xn.data_ptr(),
                                             xn.shape[0], no.data_ptr(), x_a.data_ptr(), N, L.ptr_array(ps[0]),
                                             L.ptr_array(ps[1]), 4, 64, 0.2, o.data_ptr(), 512, sv_s, sv_n, st)
            assert rc == 0, rc

        def phase(ph):
            def run(o, sv_s, sv_n):
                rc = lib.uavgnn_gatv2_hetero_fwd_phases(xs.data_ptr(), xs.shape[0], so.data_ptr(), L.ptr(order),
                                                        xn.data_ptr(), xn.shape[0], no.data_ptr(), x_a.data_ptr(), N,
                                                        L.ptr_array(ps[0]), L.ptr_array(ps[1]), 4, 64, 0.2, o.data_ptr(),
                                                        512, sv_s, sv_n, ph, st)
                assert rc == 0, rc
            return run

        def split(o, sv_s, sv_n):
            for i, (x, off, od, FS, sv) in enumerate(((xs, so, order, 4, sv_s), (xn, no, None, 2, sv_n))):
                rc = lib.uavgnn_gatv2_fwd(x.data_ptr(), x.shape[0], FS, x_a.data_ptr(), 2, off.data_ptr(), L.ptr(od), N,
                                          *[t.data_ptr() for t in ps[i]], 4, 64, 0.2, o.data_ptr() + 1024 * i, 512, sv, st)
                assert rc == 0, rc

        fused(out, a_s.data_ptr(), a_n.data_ptr())
        split(out2, a_s2.data_ptr(), a_n2.data_ptr())
        th.cuda.synchronize()
        err = float((out - out2).abs().max()) / max(float(out2.abs().max()), 1e-30)
        erra = max(float((a_s - a_s2).abs().max()) if xs.shape[0] else 0.0, float((a_n - a_n2).abs().max()))
        Es, En = xs.shape[0], xn.shape[0]
        for name, fn, sv in (("fused", fused, False), ("fused, fp32-MFMA score GEMM", phase(3 | 256), False),
                             ("  phase S only", phase(1), False), ("  phase N only", phase(2), False),
                             ("  no phase (start-up only)", phase(0), False), ("per relation (2 launches)", split, False),
                             ("fused + save", fused, True), ("per relation + save", split, True)):
            args = (out, a_s.data_ptr(), a_n.data_ptr()) if sv else (out, None, None)
            ms = time_ms(lambda: fn(*args), a.reps)
            by = 16 * Es + 8 * En + N * (8 + 8 + 2048) + (16 * (Es + En) if sv else 0)
            fl = 3360 * Es + 2320 * En + 7168 * N
            print(f"K1 fwd both relations {dist:5s} {name:28s} {ms * 1e3:8.2f} {by / ms / 1e6:9.1f} {by / ms / 1e6 / 80:6.2f} "
                  f"{fl / ms / 1e9:8.2f} {fl / ms / 1e9 / 1.573:6.2f}")
        print(f"    max rel diff fused vs per relation: out {err:.2e}  attn {erra:.2e}   (E_seen={Es}, E_near={En}, N={N})")


if __name__ == "__main__":
    main()
