"""Where the GRU cell's time goes: the operand-plane kernel (csrc/gru_x3p.hip) compiled with parts REMOVED (-DUAVGNN_X3P_DBG=bits:
1 no DMA behind the first slices, 2 no gate epilogue, 4 no MFMA, 8 no fragment reads in the loop) and timed at C3 size.
    python tools/cell_ablate.py --build     (here: hipcc cross-compiles tools/_build/x3p_dbg*.so)
    python tools/cell_ablate.py             (GPU box: us per call per variant)"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build")
VARIANTS = [0, 1, 16, 32, 2, 3, 4, 8, 9, 11, 12, 14, 14 + 16, 14 + 32, 15]
NAMES = {1: "no DMA", 16: "no activation DMA", 32: "no weight DMA", 2: "no epilogue", 4: "no MFMA", 8: "no fragment reads"}


def so(v):
    return os.path.join(OUT, f"x3p_dbg{v}.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "uav_bs_ctrl_amd", "csrc", "gru_x3p.hip")

    def one(v):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DUAVGNN_X3P_DBG={v}",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), src, "-o", so(v)], check=True)

    with ThreadPoolExecutor(8) as ex:
        list(ex.map(one, VARIANTS))


def main():
    import torch as th
    from uav_bs_ctrl_amd import _lib as L
    N, H, M = 32768, 256, 64
    dev = th.device("cuda")
    lib = L.lib()
    h = 0.5 * th.randn(N, H, device=dev)
    W_ih, W_hh = th.randn(3 * H, H + M, device=dev) / 18, th.randn(3 * H, H, device=dev) / 16
    b = th.zeros(3 * H, device=dev)
    w_tiles = th.empty(lib.uavgnn_gru_weight_tiles_bytes(H + M, H), dtype=th.uint8, device=dev)
    lib.uavgnn_gru_split_weight_tiles(W_ih.data_ptr(), H + M, W_hh.data_ptr(), H, w_tiles.data_ptr(), L.stream())
    # timing only: the planes are bf16 noise of moderate size (0x3c00.. patterns), not a real split
    planes = (th.randn(lib.uavgnn_tarmac_msg_planes_bytes(N, H, M) // 2, device=dev) * 0.3).to(th.bfloat16)
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

    print(f"# GRU cell from operand planes, N = {N}, K_in = {H + M}, H = {H}: us per call with parts of the kernel removed (opt 0 | opt 9)")
    for v in VARIANTS:
        dl = ctypes.CDLL(so(v))
        f = dl.uavgnn_gru_cell_fwd_planes_opts
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        ts = []
        for opt in (0, 9):
            call = lambda: f(planes.data_ptr(), H + M, h.data_ptr(), N, H, w_tiles.data_ptr(), b.data_ptr(), b.data_ptr(), h2.data_ptr(),  # noqa: E731
                             None, opt, L.stream())
            assert call() == 0
            ts.append(min(timeit(call) for _ in range(2)))
        what = " + ".join(n for bit, n in NAMES.items() if v & bit) or "complete kernel"
        print(f"dbg {v:2d}  {ts[0]:6.1f} | {ts[1]:6.1f}   {what}")


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
