#!/usr/bin/env python
"""Re-tune the vendor GEMM solutions of the bench workload with PyTorch's TunableOp (run on the GPU box).

    python tools/tune_gemms.py            # writes uav_bs_ctrl_amd/tuned/gemm_gfx950.csv

Runs one rollout + update cycle of the C3 / exp3 workload with tuning ON (every GEMM shape of the path is timed against
all rocBLAS / hipBLASLt solutions, operands rotated through a 512 MB buffer so that nothing is L2-resident), then the
results are written next to the package and picked up at import (uav_bs_ctrl_amd.tuned.enable_tuned_gemms)."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tmp = os.path.join(ROOT, "gpurun_out", "tunableop_new.csv")
os.makedirs(os.path.dirname(tmp), exist_ok=True)
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=tmp,
                  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="40", PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="5",
                  PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE="512")

import torch as th  # noqa: E402
import torch.cuda.tunable as tun  # noqa: E402

from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", nargs="*", default=["4096x8x80"], help="BxnxM workloads to tune (update + act at B envs)")
ap.add_argument("--merge", action="store_true", help="keep the rows of the existing csv for shapes not tuned now")
a = ap.parse_args()
for size in a.sizes:
    B, n, M = (int(v) for v in size.split("x"))
    env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=50)
    L = MultiAgentQLearner(env_info, exp3_args("cuda"))
    batch = make_sequence(B, n, M, 50, "dense" if B >= 1024 else "env", th.device("cuda"), seed=1, distinct=4)
    h = L.init_hidden(B)
    for t in range(3):
        _, h = L.act(batch["obs"][t], h, 0.05)
    L.update(batch)
    th.cuda.synchronize()
tun.write_file(tmp) if hasattr(tun, "write_file") else None
src = tmp if os.path.exists(tmp) else tmp.replace(".csv", "0.csv")
dst = os.path.join(ROOT, "uav_bs_ctrl_amd", "tuned", "gemm_gfx950.csv")
new_lines = open(src).read().splitlines()
if a.merge and os.path.exists(dst):
    have = {tuple(l.split(",")[:2]) for l in new_lines if not l.startswith("Validator")}
    new_lines += [l for l in open(dst).read().splitlines()
                  if l and not l.startswith("Validator") and tuple(l.split(",")[:2]) not in have]
open(dst, "w").write("\n".join(new_lines) + "\n")
print(open(dst).read())
