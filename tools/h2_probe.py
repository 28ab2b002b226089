#!/usr/bin/env python
"""The GRU cell at C3 size (N_a = 32 768, GRUCell(320 -> 256)) on the f16x2 arithmetic (csrc/gru_h2.hip: three f16 products per fp32
product) against the bf16x3 cell (csrc/gru_x3.hip: six bf16 products) - no-grad and with the saved pre-activations - and the fused
TarMAC message launch with and without the row maxima it hands to the f16x2 cell; error of h' against float64 for f16x2 / bf16x3 /
vendor fp32 GEMMs + gate kernel on the same data.  GPU box.

    python tools/h2_probe.py            # two passes over all arms: read the second
"""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import _lib as L, enable_tuned_gemms, ops  # noqa: E402

enable_tuned_gemms()
dev = th.device("cuda")
N, H, M, K, n = 32768, 256, 64, 16, 8
K_in = H + M
lib = L.lib()
gen = th.Generator(device=dev).manual_seed(0)
cell = th.nn.GRUCell(K_in, H).to(dev)
x = th.relu(th.randn(N, H, device=dev, generator=gen))
c = 0.5 * th.randn(N, M, device=dev, generator=gen)
h = th.tanh(th.randn(N, H, device=dev, generator=gen))
inp = th.cat((x, c), 1)
W_ih, W_hh, b_ih, b_hh = [t.detach() for t in (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]


def time_us(fn, reps=30):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


st = L.stream()
p_x3 = th.empty(lib.uavgnn_gru_cell_x3_workspace_bytes(K_in, H), dtype=th.uint8, device=dev)
L.check(lib.uavgnn_gru_split_weights(W_ih.data_ptr(), K_in, W_hh.data_ptr(), H, p_x3.data_ptr(), st), "split x3")
p_h2 = th.empty(lib.uavgnn_gru_cell_h2_workspace_bytes(K_in, H), dtype=th.uint8, device=dev)
L.check(lib.uavgnn_gru_split_weights_h2(W_ih.data_ptr(), K_in, W_hh.data_ptr(), H, p_h2.data_ptr(), st), "split h2")
rm = ops.row_absmax(x, c, h)
h2o, pre = th.empty_like(h), th.empty(N, 4 * H, device=dev)
fl = 2.0 * N * 3 * H * (K_in + H)


def x3(two_piece, save):
    a, ld, k1, b, ldb, k2 = (x, H, H, c, M, M) if two_piece else (inp, K_in, K_in, None, 0, 0)
    return lambda: L.check(lib.uavgnn_gru_cell_fwd_x3_opts(a.data_ptr(), ld, k1, L.ptr(b), ldb, k2, h.data_ptr(), N, H, p_x3.data_ptr(), b_ih.data_ptr(),
                                                           b_hh.data_ptr(), h2o.data_ptr(), pre.data_ptr() if save else None, 0, st), "x3")


def h2(two_piece, save):
    a, ld, k1, b, ldb, k2 = (x, H, H, c, M, M) if two_piece else (inp, K_in, K_in, None, 0, 0)
    return lambda: L.check(lib.uavgnn_gru_cell_fwd_h2(a.data_ptr(), ld, k1, L.ptr(b), ldb, k2, h.data_ptr(), N, H, rm.data_ptr(), p_h2.data_ptr(),
                                                      b_ih.data_ptr(), b_hh.data_ptr(), h2o.data_ptr(), pre.data_ptr() if save else None, st), "h2")


with th.no_grad():
    c64 = th.nn.GRUCell(K_in, H).to(dev).double()
    c64.load_state_dict({k: v.double() for k, v in cell.state_dict().items()})
    ref = c64(inp[:8192].double(), h[:8192].double())


def err(fn):
    fn()
    e = (h2o[:8192].double() - ref).abs()
    return f"h' vs fp64 max {e.max().item():.2e} mean {e.mean().item():.2e}"


for rnd in (1, 2):
    print(f"--- pass {rnd}")
    for name, mk in (("bf16x3 cell", x3), ("f16x2 cell ", h2)):
        t0, t1, t2 = time_us(mk(True, False)), time_us(mk(False, True)), time_us(mk(False, False))
        print(f"{name}: no-grad two-piece {t0:6.1f} us ({fl / t0 * 1e-6:5.1f} TF fp32-equivalent) | with saves {t1:6.1f} us | one piece no saves {t2:6.1f} us | {err(mk(True, False))}")
    with th.no_grad():
        ops.GRU_FUSED = ops.GEMM_X3 = False
        o = ops.gru_cell(inp[:8192], h[:8192], cell)
        ops.GRU_FUSED = ops.GEMM_X3 = True
        e = (o.double() - ref).abs()
        print(f"vendor fp32 GEMMs + gate kernel: h' vs fp64 max {e.max().item():.2e} mean {e.mean().item():.2e}")
    print(f"uavgnn_row_absmax over [x || c || h] (a pass of its own: what the shipped path does NOT pay): {time_us(lambda: ops.row_absmax(x, c, h)):6.1f} us")
    print(f"uavgnn_gru_split_weights_h2: {time_us(lambda: lib.uavgnn_gru_split_weights_h2(W_ih.data_ptr(), K_in, W_hh.data_ptr(), H, p_h2.data_ptr(), st)):5.1f} us   "
          f"uavgnn_gru_split_weights: {time_us(lambda: lib.uavgnn_gru_split_weights(W_ih.data_ptr(), K_in, W_hh.data_ptr(), H, p_x3.data_ptr(), st)):5.1f} us")
    # the message launch with and without the row maxima
    Wp = (0.1 * th.randn(M + 2 * K, 2 * H, device=dev, generator=gen)).contiguous()
    bp = 0.1 * th.randn(M + 2 * K, device=dev, generator=gen)
    tiles = th.empty(lib.uavgnn_tarmac_msg_weight_bytes(H, M, K), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_tarmac_msg_prepare(Wp.data_ptr(), 2 * H, H, M, K, tiles.data_ptr(), st), "prepare")
    off = th.arange(0, N * n + 1, n, dtype=th.int32, device=dev)
    src = ((th.arange(N, device=dev) // n * n).repeat_interleave(n) + th.arange(n, device=dev).repeat(N)).to(th.int32)
    co, rmo = th.empty(N, M, device=dev), th.empty(N, device=dev)
    a_s, proj, xc = th.empty(N * n, device=dev), th.empty(N, M + 2 * K, device=dev), th.empty(N, K_in, device=dev)
    head = (x.data_ptr(), H, h.data_ptr(), H, N, H, n, tiles.data_ptr(), bp.data_ptr(), M, K, off.data_ptr(), src.data_ptr(), 1.0 / K)
    t_a = time_us(lambda: L.check(lib.uavgnn_tarmac_msg_fwd(*head, co.data_ptr(), M, None, None, 0, None, 0, None, st), "msg"))
    t_b = time_us(lambda: L.check(lib.uavgnn_tarmac_msg_fwd_rowmax(*head, co.data_ptr(), M, None, None, 0, None, 0, rmo.data_ptr(), st), "msg rm"))
    t_c = time_us(lambda: L.check(lib.uavgnn_tarmac_msg_fwd(*head, xc.data_ptr() + 4 * H, K_in, a_s.data_ptr(), proj.data_ptr(), M + 2 * K, xc.data_ptr(), K_in, None, st), "msg tr"))
    t_d = time_us(lambda: L.check(lib.uavgnn_tarmac_msg_fwd_rowmax(*head, xc.data_ptr() + 4 * H, K_in, a_s.data_ptr(), proj.data_ptr(), M + 2 * K, xc.data_ptr(), K_in, rmo.data_ptr(), st), "msg tr rm"))
    want = th.maximum(th.maximum(x.abs().max(1).values, h.abs().max(1).values), xc[:, H:].abs().max(1).values)
    print(f"tarmac_msg_fwd: no-grad {t_a:5.1f} us, + row maxima {t_b:5.1f} us | training {t_c:5.1f} us, + row maxima {t_d:5.1f} us | row maxima exact: {bool(th.equal(rmo, want))}")

# ---- dense layers: f16x2 (csrc/gemm_h2.hip) against bf16x3 (csrc/gemm_x3.hip) at the shapes of the update's input-gradient products
print("--- GEMMs (second pass figures)")
for name, M_, K1, K2, N_, acc in (("d h += d_gh W_hh        [32768 x 768] -> 256, accumulate", 32768, 768, 0, 256, True),
                                  ("d x = [d_gi || d_proj] W [32768 x (768 + 96)] -> 256", 32768, 768, 96, 256, False),
                                  ("d K1out = dy W_aggr     [1671168 x 256] -> 512", 1671168, 256, 0, 512, False)):
    a1 = th.randn(M_, K1, device=dev, generator=gen) * 1e-3
    a2 = th.randn(M_, K2, device=dev, generator=gen) * 1e-3 if K2 else None
    W = 0.1 * th.randn(K1 + K2, N_, device=dev, generator=gen)
    y = th.zeros(M_, N_, device=dev)
    rm1 = ops.row_absmax(a1)
    rm2 = ops.row_absmax(a2) if K2 else None
    with ops.frozen_weights():
        if K2:
            f_h2 = lambda: ops.gemm_h2(a1, W[:K1], rm1, True, out=y, a2=a2, W2=W[K1:], rowmax2=rm2)
            f_x3 = lambda: ops.gemm_x3_cat(a1, a2, W[:K1], W[K1:], y)
        else:
            f_h2 = lambda: ops.gemm_h2(a1, W, rm1, True, out=y, accumulate=acc)
            f_x3 = lambda: ops.gemm_x3(a1, W, True, out=y, accumulate=acc)
        for _ in range(2):
            t_h2, t_x3 = time_us(f_h2, reps=10), time_us(f_x3, reps=10)
    fl_ = 2.0 * M_ * N_ * (K1 + K2)
    print(f"{name}: f16x2 {t_h2:8.1f} us ({fl_ / t_h2 * 1e-6:6.1f} TF) | bf16x3 {t_x3:8.1f} us ({fl_ / t_x3 * 1e-6:6.1f} TF)")
    del a1, a2, y
