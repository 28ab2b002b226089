#!/usr/bin/env python
"""Kernel micro-benchmark (GPU box): K1 forward MFMA vs VALU, K1 backward, per distribution.  Prints a small table.

    python tools/kbench.py [--B 4096] [--reps 20]
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
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--M", type=int, default=80)
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--ld", type=int, default=256)
    ap.add_argument("--dists", default="dense,env")
    a = ap.parse_args()
    dev = th.device("cuda")
    gen = th.Generator(device=dev)
    gen.manual_seed(0)
    th.manual_seed(0)
    lib = L.lib()
    st = L.stream()
    print(f"{'case':34s} {'ms':>8s} {'alg GB/s':>9s} {'%HBM':>6s} {'TFLOP/s':>8s} {'%fp32':>6s}")
    for dist in a.dists.split(","):
        hb = synth_batch_gpu(a.B, 8, a.M, dist, dev, gen)
        x_a = hb.agent_feat()
        N = x_a.shape[0]
        for et, FS in (("seen", 4), ("near", 2)):
            x_src, off = hb.relation_segments(et)
            _o = hb.relation_order(et)
            ORDER = _o.data_ptr() if (a.order and _o is not None) else None
            E = x_src.shape[0]
            conv = GATv2Conv((FS, 2), 64, 4).to(dev)
            with th.no_grad():
                for b in (conv.fc_src.bias, conv.fc_dst.bias, conv.res_fc.bias):
                    b.normal_(0, 0.1)
            p = [t.detach().contiguous() for t in (conv.fc_src.weight, conv.fc_src.bias, conv.fc_dst.weight,
                                                   conv.fc_dst.bias, conv.attn, conv.res_fc.weight, conv.res_fc.bias)]
            out = th.empty(N, a.ld, device=dev)
            out2 = th.empty(N, a.ld, device=dev)
            a_save = th.empty(max(E, 1), 4, device=dev)
            a_save2 = th.empty(max(E, 1), 4, device=dev)

            def call(fn, o, sv):
                rc = fn(x_src.data_ptr(), x_src.shape[0], FS, x_a.data_ptr(), 2, off.data_ptr(), ORDER, N, *[t.data_ptr() for t in p], 4, 64,
                        0.2, o.data_ptr(), a.ld, sv, st)
                assert rc == 0, rc

            call(lib.uavgnn_gatv2_fwd, out, a_save.data_ptr())
            call(lib.uavgnn_gatv2_fwd_valu, out2, a_save2.data_ptr())
            th.cuda.synchronize()
            err = float((out[:, :256] - out2[:, :256]).abs().max()) / max(float(out2[:, :256].abs().max()), 1e-30)
            erra = float((a_save - a_save2).abs().max())
            bytes_inf = 4 * FS * E + N * (8 + 4 + 1024)
            flops = E * (3360 if FS == 4 else 2320) + N * 3584
            # auto = what uavgnn_gatv2_fwd dispatches to (low-degree kernel for `near`, MFMA row tiles for `seen`)
            for name, fn, sv in (("auto", lib.uavgnn_gatv2_fwd, None), ("mfma", lib.uavgnn_gatv2_fwd_mfma, None),
                                 ("valu", lib.uavgnn_gatv2_fwd_valu, None),
                                 ("auto+save", lib.uavgnn_gatv2_fwd, a_save.data_ptr())):
                ms = time_ms(lambda: call(fn, out, sv), a.reps)
                by = bytes_inf + (16 * E if sv else 0)
                print(f"fwd {dist:5s} {et:4s} F={FS} {name:10s}      {ms:8.4f} {by / ms / 1e6:9.1f} "
                      f"{by / ms / 1e6 / 80:6.2f} {flops / ms / 1e9:8.2f} {flops / ms / 1e9 / 1.573:6.2f}")
            print(f"    max rel diff auto vs valu: out {err:.2e}  attn {erra:.2e}")
            # backward
            d_out = th.randn(N, a.ld, device=dev)
            g = [th.empty_like(t) for t in p]
            wsb = lib.uavgnn_gatv2_bwd_workspace_bytes(FS, 256)
            ws = th.empty(wsb // 4, device=dev)

            def bwd():
                rc = lib.uavgnn_gatv2_bwd(x_src.data_ptr(), x_src.shape[0], FS, x_a.data_ptr(), 2, off.data_ptr(), ORDER, N,
                                          *[t.data_ptr() for t in p[:5]], 4, 64, 0.2, out.data_ptr(), d_out.data_ptr(),
                                          a.ld, a_save.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                          g[3].data_ptr(), g[4].data_ptr(), g[5].data_ptr(), g[6].data_ptr(),
                                          ws.data_ptr(), wsb, st)
                assert rc == 0, rc
            call(lib.uavgnn_gatv2_fwd, out, a_save.data_ptr())
            ms = time_ms(bwd, a.reps)
            by = (4 * FS + 16) * E + N * (12 + 2048)
            print(f"bwd {dist:5s} {et:4s} F={FS}                 {ms:8.4f} {by / ms / 1e6:9.1f} "
                  f"{by / ms / 1e6 / 80:6.2f} {2 * flops / ms / 1e9:8.2f} {2 * flops / ms / 1e9 / 1.573:6.2f}")
            if FS == 2:    # A/B: the generic one-destination-per-wavefront kernel
                def bwd_generic():
                    rc = lib.uavgnn_gatv2_bwd_generic(x_src.data_ptr(), x_src.shape[0], FS, x_a.data_ptr(), 2, off.data_ptr(), ORDER, N,
                                                      *[t.data_ptr() for t in p[:5]], 4, 64, 0.2, out.data_ptr(), d_out.data_ptr(),
                                                      a.ld, a_save.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                                      g[3].data_ptr(), g[4].data_ptr(), g[5].data_ptr(), g[6].data_ptr(),
                                                      ws.data_ptr(), wsb, st)
                    assert rc == 0, rc
                ms = time_ms(bwd_generic, a.reps)
                print(f"bwd {dist:5s} {et:4s} F={FS} generic kernel  {ms:8.4f} {by / ms / 1e6:9.1f} "
                      f"{by / ms / 1e6 / 80:6.2f} {2 * flops / ms / 1e9:8.2f} {2 * flops / ms / 1e9 / 1.573:6.2f}")


if __name__ == "__main__":
    main()
