"""Times the fused TarMAC message launch (csrc/tarmac_msg.hip) at C3 size against what it replaces (two projection GEMMs +
uavgnn_talk_attn_env_fwd), per call class.  usage: python tools/msg_probe.py [B n]"""
import sys

import torch as th

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uav_bs_ctrl_amd import _lib as L, ops  # noqa: E402

B, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8)
H, M, K = 256, 64, 16
dev = th.device("cuda")
gen = th.Generator(device=dev)
gen.manual_seed(0)
g = bench.synth_batch_gpu(B, n, 80, "dense", dev, gen)
N = B * n
x, h = th.randn(N, H, device=dev), th.randn(N, H, device=dev)
Wp, bp = th.randn(M + 2 * K, 2 * H, device=dev) / 22.6, th.zeros(M + 2 * K, device=dev)
off, src = g.talk_csc()
lib = L.lib()
tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
L.check(lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), L.stream()), "prep")
E = src.shape[0]
c = th.empty(N, M, device=dev)
inp = th.empty(N, H + M, device=dev)
a_save = th.empty(E, device=dev)
proj = th.empty(N, M + 2 * K, device=dev)
planes = th.empty(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M), dtype=th.uint8, device=dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def msg(train, pl, dbg=0):
    def f():
        rc = lib.uavgnn_tarmac_msg_fwd_dbg(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off.data_ptr(),
                                           src.data_ptr(), 1.0 / K, (inp.data_ptr() + 4 * H) if train else c.data_ptr(), (H + M) if train else M,
                                           a_save.data_ptr() if train else None, proj.data_ptr() if train else None, M + 2 * K,
                                           inp.data_ptr() if train else None, H + M, planes.data_ptr() if pl else None, dbg, L.stream())
        assert rc == 0
    return f


env = ops._talk_env(g, M, K)


def old(train):
    def f():
        p = th.addmm(bp, x, Wp[:, :H].t())
        p.addmm_(h, Wp[:, H:].t())
        ld = M + 2 * K
        if train:
            ops._launch_talk_fwd(env, p.data_ptr() + 4 * M, ld, p.data_ptr() + 4 * (M + K), ld, p.data_ptr(), ld, K, M, off, src, N, 1.0 / K,
                                 inp.data_ptr() + 4 * H, H + M, a_save.data_ptr(), x.data_ptr(), H, H)
        else:
            ops._launch_talk_fwd(env, p.data_ptr() + 4 * M, ld, p.data_ptr() + 4 * (M + K), ld, p.data_ptr(), ld, K, M, off, src, N, 1.0 / K,
                                 c.data_ptr(), M, a_save.data_ptr(), None, 0, 0)
    return f


bench.enable_tuned = None
off_none = th.zeros_like(off)


def msg_noedges():
    rc = lib.uavgnn_tarmac_msg_fwd(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off_none.data_ptr(),
                                   src.data_ptr(), 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, None, L.stream())
    assert rc == 0


print(f"# fused launch on a batch WITHOUT talk edges (GEMM loop + outputs only): {timeit(msg_noedges):.1f} us")
for dbg, what in ((1, "no weight-slice traffic"), (2, "no MFMAs"), (4, "no activation loads"), (8, "no barriers"), (3, "no weights, no MFMAs"),
                  (7, "no weights / MFMAs / activation loads"), (15, "nothing but the split and the tail")):
    def f(dbg=dbg):
        assert lib.uavgnn_tarmac_msg_fwd_dbg(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off_none.data_ptr(),
                                             src.data_ptr(), 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, None, dbg, L.stream()) == 0
    print(f"#   ablation (no edges, one wavefront per row tile) dbg={dbg:2d} {what:42s}: {timeit(f):6.1f} us")
print(f"# N = {N} rows ({B} graphs of {n}), H {H}, M {M}, K {K}; us per call, 30 back-to-back calls between one event pair")
for train in (False, True):
    t_old = timeit(old(train))
    t_new = timeit(msg(train, False))
    t_one = timeit(msg(train, False, 16))
    t_pl = timeit(msg(train, True))
    rd = 2 * N * H * 4
    wr = N * M * 4 + (N * (H + M + 2 * K) * 4 + E * 4 if train else 0)
    print(f"train={int(train)}: two GEMMs + K3b {t_old:7.1f} | fused, wavefront pair per row tile {t_new:7.1f} ({(rd + wr) / t_new / 1e6:.2f} TB/s algorithmic) | "
          f"one wavefront per row tile {t_one:7.1f} | "
          f"fused + operand planes {t_pl:7.1f} ({(rd + wr + planes.numel()) / t_pl / 1e6:.2f} TB/s)")
