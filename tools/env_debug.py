import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
from tests.test_env_sim import _case
from uav_bs_ctrl_amd.sim import BatchedUbsCoverageEnv
case = "debug"
z, p, steps = _case(case)
f = lambda t, k: z[f"{case}:t{t}:{k}"]
env = BatchedUbsCoverageEnv(p, 1)
env.reset(pos_ubs=f(0, "pos_ubs")[None], pos_gts=z[f"{case}:pos_gts"][None], prior=f(0, "prior_used")[None])
print("t0", env.out["gt_ubs"], env.prior)
env.prior.copy_(th.as_tensor(f(1, "prior_used")[None]).to(th.int32))
env.step(th.as_tensor(f(1, "actions")[None]).cuda())
print("pos", env.pos_ubs, "\nd", env.out["d_u2g"], "\ngt_ubs", env.out["gt_ubs"], env.out["gt_rb"], "\nrate", env.out["rate_per_gt"], f(1, "rate_per_gt"))
print(p)
print("moves", env.moves, env.moves.is_contiguous(), env.n_actions)
a = th.as_tensor(f(1, "actions")[None]).cuda()
print("actions", a, a.dtype, a.is_contiguous(), a.stride())
env2 = BatchedUbsCoverageEnv(p, 1)
env2.reset(pos_ubs=f(0, "pos_ubs")[None], pos_gts=z[f"{case}:pos_gts"][None], prior=f(0, "prior_used")[None])
for act in ([1, 0, 0], [0, 1, 0], [2, 2, 2], [3, 3, 3]):
    before = env2.pos_ubs.clone()
    env2.step(th.tensor([act], device="cuda"))
    print(act, (env2.pos_ubs - before)[0].tolist())
