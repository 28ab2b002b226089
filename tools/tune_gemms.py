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

env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
L = MultiAgentQLearner(env_info, exp3_args("cuda"))
batch = make_sequence(4096, 8, 80, 50, "dense", th.device("cuda"), seed=1, distinct=4)
h = L.init_hidden(4096)
for t in range(3):
    _, h = L.act(batch["obs"][t], h, 0.05)
L.update(batch)
th.cuda.synchronize()
tun.write_file(tmp) if hasattr(tun, "write_file") else None
src = tmp if os.path.exists(tmp) else tmp.replace(".csv", "0.csv")
dst = os.path.join(ROOT, "uav_bs_ctrl_amd", "tuned", "gemm_gfx950.csv")
shutil.copyfile(src, dst)
print(open(dst).read())
