import cProfile, pstats, sys, os, io
sys.path.insert(0, "/root/repo")
import torch as th
import bench
from uav_bs_ctrl_amd import enable_tuned_gemms, ops
from uav_bs_ctrl_amd.learner import MultiAgentQLearner
dev = th.device("cuda", 0)
enable_tuned_gemms()
th.manual_seed(0)
n, M, T, B = 8, 80, 50, 4096
learner = MultiAgentQLearner(dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=T), bench.exp3_args(str(dev)))
batch = bench.make_sequence(B, n, M, T, "dense", dev, seed=1234, distinct=4)
ops.KERNEL_TIMER.reset(enabled=False)
def rollout():
    obs = [g.fresh() for g in batch["obs"]]
    h = learner.init_hidden(B)
    for t in range(T):
        _, h = learner.act(obs[t].fresh(), h, 0.05)
for _ in range(3): rollout()
th.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4): rollout()
pr.disable()
th.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])


def update():
    obs = [g.fresh() for g in batch["obs"]]
    fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
    return learner.update(fb)


for _ in range(2):
    update()
th.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    update()
pr.disable()
th.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("---- update (3 calls) ----")
print(s.getvalue()[:5000])
