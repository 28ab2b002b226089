"""LossQ of a replayed whole-cycle graph (graphs.GraphedCycle at BASELINE config 2) read back after every replay, with a device
synchronise behind the third: the sequence must be the eager one (0.4278 0.3857 0.3437 0.2984 0.2414 0.1484 0.1083 0.0963).
usage: python tools/gc_loss_probe.py [sync|stream|none]"""
import sys, os
sys.path.insert(0, "/root/repo")
import torch as th
import bench
from uav_bs_ctrl_amd.graphs import GraphedCycle
from uav_bs_ctrl_amd.learner import MultiAgentQLearner
dev = th.device("cuda")
n, M, T, B = 4, 40, 50, 1024
th.manual_seed(0)
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=n, episode_limit=T)
L1 = MultiAgentQLearner(env_info, bench.exp3_args("cuda"))
batch = bench.make_sequence(B, n, M, T, "dense", dev, seed=1234, distinct=4)
h_row = L1.init_hidden(1)[:1].clone()
def body():
    obs = [g.fresh() for g in batch["obs"]]
    fb = dict(batch, obs=obs, obs_all=batch["obs_all"].fresh(), obs_all_next=batch["obs_all_next"].fresh())
    h = h_row.expand(n * B, -1).contiguous()
    for t in range(T):
        _, h = L1.act(obs[t].fresh(), h, 0.05)
    return L1.update(fb)
cyc = GraphedCycle(L1, body)
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
for it in range(3):
    print("replay", it, float(cyc()["LossQ"]))
if mode == "sync":
    th.cuda.synchronize()
elif mode == "stream":
    th.cuda.current_stream().synchronize()
elif mode == "none":
    pass
print("mode", mode)
for it in range(3, 6):
    print("replay", it, float(cyc()["LossQ"]))
th.cuda.synchronize()
for it in range(6, 8):
    print("replay", it, float(cyc()["LossQ"]))
o = cyc.out
q = o["QVals"]
print("QVals[0..3]", q.flatten()[:4].tolist(), "mean", float(q.mean()), "ptr", q.data_ptr(), "loss ptr", o["LossQ"].data_ptr(), "storage offset", o["LossQ"].storage_offset(), "nbytes", o["LossQ"].untyped_storage().nbytes())
