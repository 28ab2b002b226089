"""Clock and socket power while kernels run back-to-back for ~3 s each (rocm-smi sampled from the host while the queue is full):
the GRU-cell variants of tools/cell_ablate.py, the shipped cell, the fused message kernel, K1 and a dense GEMM.
usage: python tools/cell_power.py"""
import ctypes
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402
from uav_bs_ctrl_amd import _lib as L  # noqa: E402
from tools.cell_ablate import so  # noqa: E402

N, H, M = 32768, 256, 64
dev = th.device("cuda")
lib = L.lib()
x, c, h = th.randn(N, H, device=dev), th.randn(N, M, device=dev), 0.5 * th.randn(N, H, device=dev)
W_ih, W_hh = th.randn(3 * H, H + M, device=dev) / 18, th.randn(3 * H, H, device=dev) / 16
b = th.zeros(3 * H, device=dev)
w_tiles = th.empty(lib.uavgnn_gru_weight_tiles_bytes(H + M, H), dtype=th.uint8, device=dev)
lib.uavgnn_gru_split_weight_tiles(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_tiles.data_ptr(), L.stream())
w_planes = th.empty(lib.uavgnn_gru_cell_x3_workspace_bytes(H + M, H), dtype=th.uint8, device=dev)
lib.uavgnn_gru_split_weights(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_planes.data_ptr(), L.stream())
planes = (th.randn(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M) // 2, device=dev) * 0.3).to(th.bfloat16)
h2 = th.empty(N, H, device=dev)


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout
    row = out.strip().splitlines()[-1].split(",")
    mhz = [int(m) for m in re.findall(r"\((\d+)Mhz\)", out)]
    return mhz[2], float(row[-1])          # sclk, socket power


def run(name, fn, seconds=3.0):
    for _ in range(20):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 200
    n = int(seconds * 1e6 / us)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    samples = []
    t0 = time.time()
    while time.time() - t0 < seconds - 0.6 and not e1.query():
        samples.append(smi())
    e1.synchronize()
    us_long = 1e3 * e0.elapsed_time(e1) / n
    samples = samples[1:] or samples
    clk = sorted(s[0] for s in samples)
    pw = sorted(s[1] for s in samples)
    print(f"{name:42s} {us:7.1f} us (first 200) {us_long:7.1f} us (3 s)   sclk {clk[0]}..{clk[-1]} MHz   power {pw[0]:.0f}..{pw[-1]:.0f} W   ({len(samples)} samples)", flush=True)


print("idle: sclk %d MHz, %.0f W" % smi())
NAMES = {1: "no DMA", 16: "no activation DMA", 32: "no weight DMA", 2: "no epilogue", 4: "no MFMA", 8: "no fragment reads"}
for v in (0, 1, 8, 9, 4, 11, 14):
    dl = ctypes.CDLL(so(v))
    f = dl.uavgnn_gru_cell_fwd_planes_opts
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    args = (planes.data_ptr(), H + M, h.data_ptr(), N, H, w_tiles.data_ptr(), b.data_ptr(), b.data_ptr(), h2.data_ptr(), None, 9, L.stream())
    what = " + ".join(n for bit, n in NAMES.items() if v & bit) or "complete"
    run(f"cell from planes, {what}", lambda: f(*args))
run("cell, shipped (in-kernel split)", lambda: lib.uavgnn_gru_cell_fwd_x3_cat(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), N, H, w_planes.data_ptr(),
                                                                             b.data_ptr(), b.data_ptr(), h2.data_ptr(), None, L.stream()))
A, B = th.randn(8192, 8192, device=dev, dtype=th.bfloat16), th.randn(8192, 8192, device=dev, dtype=th.bfloat16)
run("torch bf16 GEMM 8192^3 (hipBLASLt)", lambda: th.mm(A, B))
Af, Bf = th.randn(32768, 512, device=dev), th.randn(256, 512, device=dev)
Y = th.empty(32768, 256, device=dev)
run("torch fp32 GEMM 32768x512x256", lambda: th.mm(Af, Bf.t(), out=Y))
big = th.empty(1 << 28, device=dev)
run("HBM copy 1 GiB", lambda: big[: 1 << 27].copy_(big[1 << 27:]))
