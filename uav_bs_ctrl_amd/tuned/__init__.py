"""Vendor-GEMM solution selection for the path's layer shapes.

The dense half of the path (f_aggr, TarMAC projections, GRU W_ih / W_hh, Q head and their gradients) runs on the vendor
fp32 GEMM (hipBLASLt / rocBLAS through PyTorch).  The library's default heuristic is 10-30 % off the best solution it
ships for several of the backward shapes (e.g. the 16-way batched weight-gradient GEMM [768 x 2048] x [2048 x 256]:
127-140 us by default, 92 us tuned), so the solutions PyTorch's TunableOp found on an MI355X for the C3 / exp3 shapes are
recorded in ``gemm_gfx950.csv`` and selected ON REQUEST (``uav_bs_ctrl_amd.enable_tuned_gemms()``, as bench.py does, or
``UAVGNN_TUNED_GEMM=1`` in the environment: TunableOp is a process-wide PyTorch switch, so a library import must not
flip it behind the host application's back) - selection only: tuning stays OFF at run time, shapes that are
not in the file use the default heuristic, and PyTorch ignores the file when its validator lines (PyTorch / ROCm /
hipBLASLt / rocBLAS versions, gfx arch) do not match the running stack.

Re-tune:  python tools/tune_gemms.py   (on the GPU box; rewrites the csv)
Opt out of an explicit call:  UAVGNN_TUNED_GEMM=0
"""
from __future__ import annotations

import os

import torch as th

CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_gfx950.csv")


def enable_tuned_gemms() -> bool:
    """Select the recorded GEMM solutions (no tuning).  Returns True when the file was accepted."""
    if os.environ.get("UAVGNN_TUNED_GEMM", "1") == "0" or not th.cuda.is_available() or not os.path.exists(CSV):
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        return False          # the user drives TunableOp themselves (e.g. tools/tune_gemms.py): stay out of the way
    import torch.cuda.tunable as tun
    try:
        tun.enable(True)
        tun.tuning_enable(False)
        ok = bool(tun.read_file(CSV))
        if not ok:
            tun.enable(False)
        return ok
    except Exception:   # noqa: BLE001 - never let an optional speed-up break the import
        return False
