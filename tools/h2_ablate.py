"""Where the f16x2 GRU cell's time goes: csrc/gru_h2.hip compiled with parts REMOVED (-DUAVGNN_H2_DBG=bits: 1 no global loads inside the
slice loop, 2 no staging (scale + split + LDS stores), 4 no slice loop at all (prologue + epilogue), 8 no MFMAs, 16 no fragment reads
inside the loop) and timed at C3 size (N_a = 32 768, GRUCell(320 -> 256), no-grad two-piece call).  Results of the ablated builds are wrong.
    python tools/h2_ablate.py --build     (here: hipcc cross-compiles tools/_build/h2_dbg*.so)
    python tools/h2_ablate.py             (GPU box: us per call per variant)"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build")
VARIANTS = [0, 1, 2, 3, 4, 8, 16, 8 + 16, 1 + 2 + 16, 1 + 2 + 8, 1 + 2 + 8 + 16]
NAMES = {1: "no global loads", 2: "no staging", 4: "no slice loop", 8: "no MFMA", 16: "no fragment reads"}


def so(v):
    return os.path.join(OUT, f"h2_dbg{v}.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "uav_bs_ctrl_amd", "csrc", "gru_h2.hip")

    def one(v):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DUAVGNN_H2_DBG={v}",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), src, "-o", so(v)], check=True)

    with ThreadPoolExecutor(8) as ex:
        list(ex.map(one, VARIANTS))


AB = {"persistent": ["-DUAVGNN_H2_PERSIST=1"], "one_tile_per_wg": ["-DUAVGNN_H2_PERSIST=0"]}


def build_ab():
    """A/B builds: the experimental kernel of tools/ubench/gru_h2_persistent.hip (persistent grid / one workgroup per tile) and `base`, the
    shipped csrc/gru_h2.hip."""
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "tools", "ubench", "gru_h2_persistent.hip")
    jobs = [(n, src, f) for n, f in AB.items()] + [("base", os.path.join(ROOT, "uav_bs_ctrl_amd", "csrc", "gru_h2.hip"), []),
                                                   ("w4_64x64", os.path.join(ROOT, "tools", "ubench", "gru_h2_w4.hip"), [])]
    for n, f, flags in jobs:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *flags,
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "uav_bs_ctrl_amd", "csrc"), f, "-o",
                        os.path.join(OUT, f"h2_ab_{n}.so")], check=True)


def main_ab():
    import torch as th
    from uav_bs_ctrl_amd import _lib as L
    from uav_bs_ctrl_amd import ops
    N, H, M = 32768, 256, 64
    dev = th.device("cuda")
    lib = L.lib()
    th.manual_seed(3)
    x, c, h = th.relu(th.randn(N, H, device=dev)), 0.5 * th.randn(N, M, device=dev), th.tanh(th.randn(N, H, device=dev))
    W_ih, W_hh = th.randn(3 * H, H + M, device=dev) / 18, th.randn(3 * H, H, device=dev) / 16
    b = 0.1 * th.randn(3 * H, device=dev)
    planes = th.empty(lib.uavgnn_gru_cell_h2_workspace_bytes(H + M, H), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_gru_split_weights_h2(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, planes.data_ptr(), L.stream()), "split")
    rm = ops.row_absmax(x, c, h)
    names = [n for n in list(AB) + ["base", "w4_64x64"] if os.path.exists(os.path.join(OUT, f"h2_ab_{n}.so"))]
    outs = {}

    def timeit(fn, reps=40):
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

    print(f"# f16x2 GRU cell, N = {N}, K_in = {H + M}, H = {H}, two-piece call: us per call (min of 3 x 40 back-to-back calls), 2 passes")
    fns = {}
    for n in names:
        dl = ctypes.CDLL(os.path.join(OUT, f"h2_ab_{n}.so"))
        f = dl.uavgnn_gru_cell_fwd_h2
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p]
        fns[n] = f
    for ps in range(2):
        for n in names:
            f = fns[n]
            h2, pre = th.empty(N, H, device=dev), th.empty(N, 4 * H, device=dev)
            call = lambda pre_=None: f(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), N, H, rm.data_ptr(), planes.data_ptr(),  # noqa: E731
                                       b.data_ptr(), b.data_ptr(), h2.data_ptr(), pre_, L.stream())
            assert call() == 0
            t0 = min(timeit(call) for _ in range(3))
            t1 = min(timeit(lambda: call(pre.data_ptr())) for _ in range(3))
            th.cuda.synchronize()
            outs[n] = (h2.clone(), pre.clone())
            # ragged row count (not a multiple of 128, padding tiles in the last group of row blocks)
            Nr = 32768 - 128 * 5 - 37
            h3 = th.zeros(N, H, device=dev)
            assert f(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), Nr, H, rm.data_ptr(), planes.data_ptr(), b.data_ptr(), b.data_ptr(),
                     h3.data_ptr(), None, L.stream()) == 0
            th.cuda.synchronize()
            ok = bool(th.equal(h3[:Nr], outs[n][0][:Nr])) and float(h3[Nr:].abs().max()) == 0.0
            print(f"pass {ps + 1}  {n:16s} no-grad {t0:6.1f}   with saves {t1:6.1f}   ragged N ok: {ok}")
    ref = outs[names[0]]
    for n in names[1:]:
        print(f"{n} == {names[0]}: h' {bool(th.equal(outs[n][0], ref[0]))}, pre {bool(th.equal(outs[n][1], ref[1]))}")


def main():
    import torch as th
    from uav_bs_ctrl_amd import _lib as L
    from uav_bs_ctrl_amd import ops
    N, H, M = 32768, 256, 64
    dev = th.device("cuda")
    lib = L.lib()
    x, c, h = th.relu(th.randn(N, H, device=dev)), 0.5 * th.randn(N, M, device=dev), th.tanh(th.randn(N, H, device=dev))
    W_ih, W_hh = th.randn(3 * H, H + M, device=dev) / 18, th.randn(3 * H, H, device=dev) / 16
    b = th.zeros(3 * H, device=dev)
    planes = th.empty(lib.uavgnn_gru_cell_h2_workspace_bytes(H + M, H), dtype=th.uint8, device=dev)
    L.check(lib.uavgnn_gru_split_weights_h2(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, planes.data_ptr(), L.stream()), "split")
    rm = ops.row_absmax(x, c, h)
    h2 = th.empty(N, H, device=dev)

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

    print(f"# f16x2 GRU cell, N = {N}, K_in = {H + M}, H = {H}, no-grad two-piece call: us per call with parts of the kernel removed")
    for v in VARIANTS:
        dl = ctypes.CDLL(so(v))
        f = dl.uavgnn_gru_cell_fwd_h2
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p]
        call = lambda: f(x.data_ptr(), H, H, c.data_ptr(), M, M, h.data_ptr(), N, H, rm.data_ptr(), planes.data_ptr(), b.data_ptr(),  # noqa: E731
                         b.data_ptr(), h2.data_ptr(), None, L.stream())
        assert call() == 0
        t = min(timeit(call) for _ in range(3))
        what = " + ".join(n for bit, n in NAMES.items() if v & bit) or "complete kernel"
        print(f"dbg {v:2d}  {t:6.1f}   {what}")


if __name__ == "__main__":
    if "--ab" in sys.argv:
        build_ab() if "--build" in sys.argv else main_ab()
    else:
        build() if "--build" in sys.argv else main()
