#!/usr/bin/env python
"""A few launches of the fused K4 cell at C3 size for rocprofv3 PMC passes (tools/pmc.sh)."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uav_bs_ctrl_amd import ops  # noqa: E402

dev = th.device("cuda")
N, K, H = 32768, 320, 256
cell = th.nn.GRUCell(K, H).to(dev)
inp, h = th.randn(N, K, device=dev), th.randn(N, H, device=dev)
ops.GRU_X3 = "--f32" not in sys.argv
with th.no_grad():
    for _ in range(10):
        ops.gru_cell(inp, h, cell)
th.cuda.synchronize()
