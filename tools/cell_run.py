"""A few launches of every GRU cell kernel at C3 size, for rocprofv3 counter passes (tools/pmc.sh):
    bash tools/pmc.sh <out> "gru_cell" -- python tools/cell_run.py"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uav_bs_ctrl_amd import _lib as L  # noqa: E402

N, H, M, K, n = 32768, 256, 64, 16, 8
dev = th.device("cuda")
lib = L.lib()
x, c, h = th.randn(N, H, device=dev), th.randn(N, M, device=dev), 0.5 * th.randn(N, H, device=dev)
W_ih, W_hh = th.randn(3 * H, H + M, device=dev) / 18, th.randn(3 * H, H, device=dev) / 16
b_ih, b_hh = th.zeros(3 * H, device=dev), th.zeros(3 * H, device=dev)
w_planes = th.empty(lib.uavgnn_gru_cell_x3_workspace_bytes(H + M, H), dtype=th.uint8, device=dev)
lib.uavgnn_gru_split_weights(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_planes.data_ptr(), L.stream())
w_tiles = th.empty(lib.uavgnn_gru_weight_tiles_bytes(H + M, H), dtype=th.uint8, device=dev)
lib.uavgnn_gru_split_weight_tiles(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_tiles.data_ptr(), L.stream())
off = th.arange(0, N * n + 1, n, dtype=th.int32, device=dev)
src = (th.arange(N, device=dev) // n * n).repeat_interleave(n).to(th.int32) + th.arange(n, device=dev, dtype=th.int32).repeat(N)
Wp, bp = th.randn(M + 2 * K, 2 * H, device=dev) / 22, th.zeros(M + 2 * K, device=dev)
tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), L.stream())
planes = th.empty(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M), dtype=th.uint8, device=dev)
assert lib.uavgnn_tarmac_msg_fwd(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off.data_ptr(), src.data_ptr(),
                                 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, planes.data_ptr(), L.stream()) == 0
h2 = th.empty(N, H, device=dev)
for _ in range(6):
    lib.uavgnn_gru_cell_fwd_x3_cat(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), N, H, w_planes.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                                   h2.data_ptr(), None, L.stream())
    for opt in (0, 5):
        lib.uavgnn_gru_cell_fwd_planes_opts(planes.data_ptr(), H + M, h.data_ptr(), N, H, w_tiles.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                                            h2.data_ptr(), None, opt, L.stream())
th.cuda.synchronize()
