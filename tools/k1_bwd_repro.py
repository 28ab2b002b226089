#!/usr/bin/env python
"""Repeatability and A/B probe of the K1 backward (`seen` shape: F_src 4, nh 4, D 64) through the C-ABI.

    CASES="[(4096,64,64),(8192,16,130)]" REPS=100 python tools/k1_bwd_repro.py       launches per case that differ from the first
    REL=1  ... per-gradient difference between uavgnn_gatv2_bwd (matrix cores for mean degree >= 16) and the generic kernel
    ROWS=1 ... which workgroups' partial rows differ between launches, and in which gradient segment
    TIME=1 ... 20 extra launches of each entry (for rocprofv3 --kernel-trace)
    UAVGNN_K1_BWD_MFMA=0 sends every destination through the packed-FMA kernel.

This is the script that located the gfx950 packed-fp32 / bf16-MFMA hazard (tools/ubench/mfma_pk_hazard.hip):
CASES are (destinations, min degree, max degree).
"""
import sys, os
sys.path.insert(0, "/root/repo")
import torch as th
from uav_bs_ctrl_amd import _lib as L
def case(N, lo, hi, seed):
    gen = th.Generator().manual_seed(seed)
    deg = th.randint(lo, hi + 1, (N,), generator=gen)
    off = th.zeros(N + 1, dtype=th.int32); off[1:] = th.cumsum(deg, 0); E = int(off[-1]); dev = "cuda"
    x_src = (th.rand(max(E, 1), 4, generator=gen) * 2 - 1).to(dev); x_dst = th.rand(N, 2, generator=gen).to(dev)
    H = 256
    prm = [(0.5 * th.randn(s, generator=gen)).to(dev) for s in ((H, 4), (H,), (H, 2), (H,), (H,), (H, 2), (H,))]
    out = th.empty(N, 512, device=dev); a_save = th.empty(max(E, 1), 4, device=dev)
    lib, st, offd = L.lib(), L.stream(), off.to(dev)
    assert lib.uavgnn_gatv2_fwd(x_src.data_ptr(), E, 4, x_dst.data_ptr(), 2, offd.data_ptr(), None, N, *[t.data_ptr() for t in prm], 4, 64, 0.2, out.data_ptr(), 512, a_save.data_ptr(), st) == 0
    d_out = th.randn(N, 512, generator=gen).to(dev)
    mode = os.environ.get("MASK", "")
    if mode == "first": d_out[2048:] = 0
    if mode == "second": d_out[:2048] = 0
    wsb = lib.uavgnn_gatv2_bwd_workspace_bytes(4, H); ws = th.empty(wsb // 4, device=dev)
    def run(fn):
        g = [th.full_like(t, float("nan")) for t in prm]
        assert fn(x_src.data_ptr(), E, 4, x_dst.data_ptr(), 2, offd.data_ptr(), None, N, *[t.data_ptr() for t in prm[:5]], 4, 64, 0.2, out.data_ptr(), d_out.data_ptr(), 512, a_save.data_ptr(), *[t.data_ptr() for t in g], ws.data_ptr(), wsb, st) == 0
        th.cuda.synchronize(); return g
    return run, lib, ws
cases = eval(os.environ.get("CASES", "[(4096, 64, 64), (4096, 48, 48), (8192, 100, 128)]"))
for (N, lo, hi) in cases:
    run, lib, ws = case(N, lo, hi, 11)
    ref = run(lib.uavgnn_gatv2_bwd)
    gen_ = run(lib.uavgnn_gatv2_bwd_generic)
    nbad = 0
    for rep in range(int(os.environ.get("REPS", "6"))):
        b = run(lib.uavgnn_gatv2_bwd)
        if any(not th.equal(x, y) for x, y in zip(ref, b)):
            nbad += 1
    print(N, lo, hi, "nondeterministic runs:", nbad, "of", os.environ.get("REPS", "6"), "; max |mf - generic| db_s", float((ref[1] - gen_[1]).abs().max()))
    if os.environ.get("TIME"):
        for fn in (lib.uavgnn_gatv2_bwd, lib.uavgnn_gatv2_bwd_generic):
            for _ in range(20): run(fn)
    if os.environ.get("ROWS"):
        g = min((N + 3) // 4, 512); P = 256 * 12    # the library's grid of the matrix-core launch
        run(lib.uavgnn_gatv2_bwd); r0 = ws[: 2 * g * P].clone().view(2 * g, P)
        names = [("dW_s", 0, 1024), ("db_s", 1024, 1280), ("dW_d", 1280, 1792), ("db_d", 1792, 2048), ("datt", 2048, 2304), ("dW_r", 2304, 2816), ("db_r", 2816, 3072)]
        for rep in range(5):
            run(lib.uavgnn_gatv2_bwd); r1 = ws[: 2 * g * P].view(2 * g, P)
            bad = (r0 != r1)
            rows = bad.any(1).nonzero().flatten().tolist()
            print(" rep", rep, "rows differing:", rows[:12], "n", len(rows))
            for r in rows[:3]:
                for nm, a, b in names:
                    idx = bad[r, a:b].nonzero().flatten()
                    if len(idx):
                        ch = idx.tolist()
                        print("    row", r, nm, "count", len(ch), "idx", ch[:24], "maxdiff", float((r0[r, a:b] - r1[r, a:b]).abs().max()))
    if os.environ.get("REL"):
        a = run(lib.uavgnn_gatv2_bwd); b = run(lib.uavgnn_gatv2_bwd_generic)
        for x, y, nm in zip(a, b, ["dW_s", "db_s", "dW_d", "db_d", "dattn", "dW_r", "db_r"]):
            d = (x - y).abs()
            print("   ", nm, "max|d|", float(d.max()), "max|ref|", float(y.abs().max()), "rel-to-max", float(d.max() / y.abs().max()), "n(|d| > 1e-5 max)", int((d > 1e-5 * y.abs().max()).sum()), "of", d.numel())
