#!/usr/bin/env python
"""Time-batched weight-gradient shapes of a C3 update on csrc/gemm_tn_x3.hip: dW[Mo, Ko] = dY^T X over (T + 1) N_a = 1 671 168
rows (the staged BPTT sequence, ops.WeightGradSink.end_sequence) against the vendor's batched split-K fp32 GEMM.

    python tools/gemm_tn_big_probe.py [--rows 1671168]
"""
import argparse
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uav_bs_ctrl_amd import _lib as L, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=51 * 32768)
a = ap.parse_args()
dev = th.device("cuda")
R = a.rows


def timed(fn, reps=5):
    fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, Mo, Ko in (("f_aggr", 256, 512), ("W_ih", 768, 320), ("W_hh", 768, 256), ("Wp_x", 96, 256), ("W_out", 9, 256)):
    dy = th.randn(R, Mo, device=dev)
    x = th.randn(R, Ko, device=dev)
    S = L.lib().uavgnn_gemm_tn_x3_chunks(R, Mo, Ko)
    t_x3 = timed(lambda: ops.gemm_tn_x3(dy, x).sum(0))
    Sv = 64
    while R % Sv:
        Sv //= 2
    t_v = timed(lambda: th.bmm(dy.view(Sv, R // Sv, Mo).transpose(1, 2), x.view(Sv, R // Sv, Ko)).sum(0))
    fl = 2.0 * Mo * Ko * R
    print(f"{name:7s} [{Mo:3d} x {Ko:3d}] rows {R}: bf16x3 S={S:3d} {t_x3:7.3f} ms = {fl / t_x3 / 1e9:6.1f} TF   "
          f"vendor split-K (S={Sv}) {t_v:7.3f} ms = {fl / t_v / 1e9:6.1f} TF")
    del dy, x
