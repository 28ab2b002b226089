#!/usr/bin/env python
"""Weight-gradient products of a C3 update (dW = dY^T X over 1 671 168 time-batched rows) on the f16x2 kernel (csrc/gemm_tn_h2.hip: LDS
transposing reads, three f16 products per fp32 product) against the vendor's fp32 split-K batched GEMM (what rounds 2-5 ran) and
csrc/gemm_tn_x3.hip: time per product, fp32-equivalent TFLOP/s, error against float64 on a 65 536-row prefix.  GPU box.
    python tools/gemm_tn_h2_probe.py [rows]"""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import _lib as L, enable_tuned_gemms  # noqa: E402

enable_tuned_gemms()
dev = th.device("cuda")
lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 51 * 32768
gen = th.Generator(device=dev).manual_seed(0)


def time_us(fn, reps=5):
    for _ in range(2):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def col_absmax(x):
    out = th.empty(x.shape[1], device=dev)
    L.check(lib.uavgnn_col_absmax(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), L.stream()), "col_absmax")
    return out


def tn_h2(dy, x, cy, cx, S=None):
    Mo, Ko = dy.shape[1], x.shape[1]
    S = S or lib.uavgnn_gemm_tn_h2_chunks(dy.shape[0], Mo, Ko)
    part = th.empty(S, Mo, Ko, device=dev)
    L.check(lib.uavgnn_gemm_tn_h2(dy.data_ptr(), dy.stride(0), Mo, x.data_ptr(), x.stride(0), Ko, dy.shape[0], cy.data_ptr(), cx.data_ptr(),
                                  part.data_ptr(), S, 0, L.stream()), "gemm_tn_h2")
    return part


def vendor(dy, x):
    S, nn = 64, dy.shape[0]
    while nn % S:
        S //= 2
    return th.bmm(dy.view(S, nn // S, -1).transpose(1, 2), x.view(S, nn // S, -1))


def tn_x3(dy, x):
    S = lib.uavgnn_gemm_tn_x3_chunks(dy.shape[0], dy.shape[1], x.shape[1])
    part = th.empty(S, dy.shape[1], x.shape[1], device=dev)
    L.check(lib.uavgnn_gemm_tn_x3(dy.data_ptr(), dy.stride(0), dy.shape[1], x.data_ptr(), x.stride(0), x.shape[1], dy.shape[0], part.data_ptr(), S, 0,
                                  L.stream()), "gemm_tn_x3")
    return part


print(f"# dW = dY^T X over {n} rows; error = max / mean |dW - fp64| / sum_n |dy x| on the first 65 536 rows")
for name, Mo, Ko in (("dW_ih  [768 x 320]", 768, 320), ("dW_hh  [768 x 256]", 768, 256), ("dW_aggr [256 x 512]", 256, 512), ("dWp_x  [96 x 256]", 96, 256)):
    # gradient-like dY: rows whose magnitudes span four orders (time steps), activation-like X
    dy = th.randn(n, Mo, device=dev, generator=gen) * th.exp2(th.randint(-14, 1, (n, 1), device=dev, generator=gen).float()) * 1e-3
    x = th.relu(th.randn(n, Ko, device=dev, generator=gen))
    cy, cx = col_absmax(dy), col_absmax(x)
    assert th.equal(cy, dy.abs().max(0).values) and th.equal(cx, x.abs().max(0).values), "column maxima"
    m = 65536
    ref = dy[:m].double().t() @ x[:m].double()
    den = dy[:m].double().abs().t() @ x[:m].double().abs()
    rows = []
    for tag, f in (("f16x2", lambda a, b: tn_h2(a, b, cy, cx).sum(0)), ("vendor fp32 split-K", lambda a, b: vendor(a, b).sum(0)),
                   ("bf16x3", lambda a, b: tn_x3(a, b).sum(0))):
        e = (f(dy[:m], x[:m]).double() - ref).abs() / den
        rows.append(f"{tag}: max {e.max().item():.2e} mean {e.mean().item():.2e}")
    fl = 2.0 * n * Mo * Ko
    t_h2 = time_us(lambda: tn_h2(dy, x, cy, cx))
    t_v = time_us(lambda: vendor(dy, x))
    t_x3 = time_us(lambda: tn_x3(dy, x))
    t_cm = time_us(lambda: (col_absmax(dy), col_absmax(x)))
    print(f"{name}: f16x2 {t_h2:8.1f} us ({fl / t_h2 * 1e-6:6.1f} TF) | vendor {t_v:8.1f} us ({fl / t_v * 1e-6:6.1f} TF) | bf16x3 {t_x3:8.1f} us "
          f"({fl / t_x3 * 1e-6:6.1f} TF) | column maxima of both operands (a pass of their own) {t_cm:7.1f} us")
    print("    " + " | ".join(rows))
    if Mo < 128:      # a narrow dY: the product the shipped path runs is the TRANSPOSE, X^T dY (ops.WeightGradSink.end_sequence): one tile 3/4 full
        t_sw = time_us(lambda: tn_h2(x, dy, cx, cy))
        e = (tn_h2(x[:m], dy[:m], cx, cy).sum(0).t().double() - ref).abs() / den
        print(f"    operands swapped (X^T dY = dW^T [{Ko} x {Mo}]): f16x2 {t_sw:8.1f} us ({fl / t_sw * 1e-6:6.1f} TF)   max {e.max().item():.2e} mean {e.mean().item():.2e}")
    del dy, x
