#!/usr/bin/env python
"""A few launches of the round-6 f16x2 kernels at C3 size for the rocprofv3 --pmc passes of tools/pmc.sh: the GRU cell (no-grad two-piece
call), the fused message launch with row maxima, the two input-gradient GEMMs of a BPTT step and the dW_hh weight gradient over 262 144
rows."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import _lib as L, ops  # noqa: E402

dev = th.device("cuda")
lib, st = L.lib(), L.stream()
N, H, M, K, n = 32768, 256, 64, 16, 8
gen = th.Generator(device=dev).manual_seed(0)
x, c, h = th.relu(th.randn(N, H, device=dev, generator=gen)), 0.5 * th.randn(N, M, device=dev, generator=gen), th.tanh(th.randn(N, H, device=dev, generator=gen))
cell = th.nn.GRUCell(H + M, H).to(dev)
Wp, bp = (0.1 * th.randn(M + 2 * K, 2 * H, device=dev, generator=gen)).contiguous(), 0.1 * th.randn(M + 2 * K, device=dev, generator=gen)
tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
L.check(lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), st), "prepare")
off = th.arange(0, N * n + 1, n, dtype=th.int32, device=dev)
src = ((th.arange(N, device=dev) // n * n).repeat_interleave(n) + th.arange(n, device=dev).repeat(N)).to(th.int32)
rm = th.empty(N, device=dev)
d_g = th.randn(N, 3 * H, device=dev, generator=gen) * 1e-3
d_p = th.randn(N, M + 2 * K, device=dev, generator=gen) * 1e-3
rg, rp = ops.row_absmax(d_g), ops.row_absmax(d_p)
dh = th.zeros(N, H, device=dev)
nb = 1 << 18
dy, xb = th.randn(nb, 3 * H, device=dev, generator=gen) * 1e-3, th.tanh(th.randn(nb, H, device=dev, generator=gen))
sink = ops.WeightGradSink()
with th.no_grad(), ops.frozen_weights():
    for _ in range(3):
        L.check(lib.uavgnn_tarmac_msg_fwd_rowmax(x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off.data_ptr(), src.data_ptr(),
                                                 1.0 / K, c.data_ptr(), M, None, None, 0, None, 0, rm.data_ptr(), st), "msg")
        ops._gru_cell_launch(x, h, cell.weight_ih, cell.bias_ih, cell.weight_hh, cell.bias_hh, save=False, inp2=c, rowmax=rm)
        ops.gemm_h2(d_g, cell.weight_hh, rg, True, out=dh, accumulate=True)
        ops.gemm_h2(d_g, cell.weight_ih[:, :H], rg, True, a2=d_p, W2=Wp[:, :H], rowmax2=rp)
        sink.weight_h2(("probe", 0), dy, xb, dy.abs().max(), xb.abs().max(), None)
th.cuda.synchronize()
print("ok")
