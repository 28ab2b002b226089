"""GRU cell at C3 size: in-kernel split (csrc/gru_x3.hip) vs prepared operand planes (csrc/gru_x3p.hip), no-grad and with saves.
usage: python tools/cell_probe.py"""
import sys

import torch as th

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uav_bs_ctrl_amd import _lib as L  # noqa: E402

N, H, M = 32768, 256, 64
dev = th.device("cuda")
lib = L.lib()
x, c, h = th.randn(N, H, device=dev), th.randn(N, M, device=dev), 0.5 * th.randn(N, H, device=dev)
W_ih, W_hh = th.randn(3 * H, H + M, device=dev) / 18, th.randn(3 * H, H, device=dev) / 16
b_ih, b_hh = th.zeros(3 * H, device=dev), th.zeros(3 * H, device=dev)
w_planes = th.empty(lib.uavgnn_gru_cell_x3_workspace_bytes(H + M, H), dtype=th.uint8, device=dev)
lib.uavgnn_gru_split_weights(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_planes.data_ptr(), L.stream())
w_tiles = th.empty(lib.uavgnn_gru_weight_tiles_bytes(H + M, H), dtype=th.uint8, device=dev)
lib.uavgnn_gru_split_weight_tiles(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_tiles.data_ptr(), L.stream())
# planes of random bf16-exact garbage are fine for timing: use the message kernel to make real ones
K, n = 16, 8
off = th.arange(0, N * n + 1, n, dtype=th.int32, device=dev)
src = (th.arange(N, device=dev) // n * n).repeat_interleave(n).to(th.int32) + th.arange(n, device=dev, dtype=th.int32).repeat(N)
Wp, bp = th.randn(M + 2 * K, 2 * H, device=dev) / 22, th.zeros(M + 2 * K, device=dev)
tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), L.stream())
planes = th.empty(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M), dtype=th.uint8, device=dev)
assert lib.uavgnn_tarmac_msg_fwd(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off.data_ptr(), src.data_ptr(),
                                 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, planes.data_ptr(), L.stream()) == 0
h2 = th.empty(N, H, device=dev)
pre = th.empty(N, 4 * H, device=dev)


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


fl = 2.0 * N * 3 * H * (H + M + H)
print(f"# GRU cell, N = {N}, K_in = {H + M}, H = {H}; us per call (30 back-to-back), fp32-equivalent TFLOP/s, x 6 / 2500 = fraction of the bf16 roof")
for save in (False, True):
    p = pre.data_ptr() if save else None
    for rep in range(2):
        t_a = timeit(lambda: lib.uavgnn_gru_cell_fwd_x3_cat(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), N, H, w_planes.data_ptr(),
                                                            b_ih.data_ptr(), b_hh.data_ptr(), h2.data_ptr(), p, L.stream()))
        line = f"save={int(save)} pass {rep}: in-kernel split {t_a:6.1f} us ({6 * fl / t_a / 1e6 / 2500:.3f}) | operand planes, opt 0..7, 8, 9, 12, 13:"
        for opt in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13):
            t_b = timeit(lambda: lib.uavgnn_gru_cell_fwd_planes_opts(planes.data_ptr(), H + M, h.data_ptr(), N, H, w_tiles.data_ptr(),
                                                                     b_ih.data_ptr(), b_hh.data_ptr(), h2.data_ptr(), p, opt, L.stream()))
            line += f" {t_b:6.1f}"
        print(line)
