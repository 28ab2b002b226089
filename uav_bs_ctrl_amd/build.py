"""Builds ``libuavgnn.so`` (the C-ABI of include/uavgnn.h) for gfx950 with hipcc.  No GPU is needed to compile.

    python -m uav_bs_ctrl_amd.build [--force]
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libuavgnn.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    cmd = [HIPCC, *FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *sources(), "-o", OUT]
    if verbose:
        print("[uav_bs_ctrl_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(OUT)
