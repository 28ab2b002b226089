#!/usr/bin/env python
"""K1-seen launch time inside the rollout loop (HIP events of ops.KERNEL_TIMER): before any update, and after one."""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp3_args, make_sequence  # noqa: E402
from uav_bs_ctrl_amd import ops  # noqa: E402
from uav_bs_ctrl_amd.learner import MultiAgentQLearner  # noqa: E402

dev = th.device("cuda")
env_info = dict(obs_shape=dict(agent=2, ubs=2, gt=4), n_actions=9, n_agents=8, episode_limit=50)
L = MultiAgentQLearner(env_info, exp3_args("cuda"))
batch = make_sequence(4096, 8, 80, 50, "dense", dev, seed=1, distinct=4)


def rollout(tag):
    ops.KERNEL_TIMER.reset(enabled=True)
    h = L.init_hidden(4096)
    for t in range(50):
        _, h = L.act(batch["obs"][t].fresh(), h, 0.05)
    s = ops.KERNEL_TIMER.summary()
    ops.KERNEL_TIMER.enabled = False
    print(tag, {k: round(v["avg_ms"] * 1e3, 1) for k, v in s.items()})


rollout("warm-up      ")
rollout("before update")
L.update(batch)
th.cuda.synchronize()
rollout("after update ")
rollout("again        ")


def loop(tag, fn, n=50):
    ops.KERNEL_TIMER.reset(enabled=True)
    with th.no_grad():
        for t in range(n):
            fn(t)
    s = ops.KERNEL_TIMER.summary()
    ops.KERNEL_TIMER.enabled = False
    print(tag, {k: round(v["avg_ms"] * 1e3, 1) for k, v in s.items()})


net = L.policy_net
obs = batch["obs"]
loop("encode only (same graph, cached order)  ", lambda t: net.encode(obs[0]))
loop("encode only (rotating graphs, cached)   ", lambda t: net.encode(obs[t % 4]))
loop("encode only (fresh graph objects)       ", lambda t: net.encode(obs[t % 4].fresh()))
h0 = L.init_hidden(4096)
xx = net.encode(obs[0]).detach()
loop("step only                               ", lambda t: net.step(obs[t % 4], xx, h0))
loop("encode + step                           ", lambda t: net.step(obs[t % 4], net.encode(obs[t % 4]), h0))

from uav_bs_ctrl_amd import _lib as LL  # noqa: E402

g0 = obs[0]
x_src, off = g0.relation_segments("seen")
x_a, N = g0.agent_feat(), g0.num_nodes("agent")
conv = net.enc.f_conv["seen"]
P = [t.detach().contiguous() for t in (conv.fc_src.weight, conv.fc_src.bias, conv.fc_dst.weight, conv.fc_dst.bias,
                                       conv.attn, conv.res_fc.weight, conv.res_fc.bias)]


def raw(ld, outbuf, params):
    ev = []
    for i in range(40):
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        rc = LL.lib().uavgnn_gatv2_fwd(x_src.data_ptr(), x_src.shape[0], 4, x_a.data_ptr(), 2, off.data_ptr(), None, N,
                                       *[t.data_ptr() for t in params], 4, 64, 0.2, outbuf.data_ptr(), ld, None, LL.stream())
        b.record()
        ev.append((a, b))
    th.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


out512 = th.empty(N, 512, device=dev)
print("raw C call, learner's weights, ld 512:", round(raw(512, out512, P), 1), "us")
th.manual_seed(0)
from uav_bs_ctrl_amd.agents.gnn_agents import GATv2Conv  # noqa: E402
c2 = GATv2Conv((4, 2), 64, 4).cuda()
P2 = [t.detach().contiguous() for t in (c2.fc_src.weight, c2.fc_src.bias, c2.fc_dst.weight, c2.fc_dst.bias, c2.attn,
                                        c2.res_fc.weight, c2.res_fc.bias)]
print("raw C call, fresh default-init weights:", round(raw(512, out512, P2), 1), "us")
print("x_a stats", float(x_a.abs().max()), "x_src", float(x_src.abs().max()), [float(t.abs().max()) for t in P])

xn, offn = g0.relation_segments("near")
convn = net.enc.f_conv["near"]
Pn = [t.detach().contiguous() for t in (convn.fc_src.weight, convn.fc_src.bias, convn.fc_dst.weight, convn.fc_dst.bias,
                                        convn.attn, convn.res_fc.weight, convn.res_fc.bias)]
Wagg, bagg = net.enc.f_aggr[0].weight if hasattr(net.enc.f_aggr, "__getitem__") else net.enc.f_aggr.weight, None
graphs = [obs[i] for i in range(4)]
segs = [(gg.relation_segments("seen"), gg.relation_segments("near"), gg.agent_feat()) for gg in graphs]


def seq(with_near, with_gemm, rotate, fresh_out):
    ev = []
    o = out512
    for i in range(40):
        (xs_, of_), (xn_, ofn_), xa_ = segs[i % 4] if rotate else segs[0]
        if fresh_out:
            o = th.empty(N, 512, device=dev)
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        LL.lib().uavgnn_gatv2_fwd(xs_.data_ptr(), xs_.shape[0], 4, xa_.data_ptr(), 2, of_.data_ptr(), None, N,
                                  *[t.data_ptr() for t in P], 4, 64, 0.2, o.data_ptr(), 512, None, LL.stream())
        b.record()
        ev.append((a, b))
        if with_near:
            LL.lib().uavgnn_gatv2_fwd(xn_.data_ptr(), xn_.shape[0], 2, xa_.data_ptr(), 2, ofn_.data_ptr(), None, N,
                                      *[t.data_ptr() for t in Pn], 4, 64, 0.2, o.data_ptr() + 1024, 512, None, LL.stream())
        if with_gemm:
            th.mm(o, Wagg.t())
    th.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return round(t[len(t) // 2], 1)


print("seen only                      ", seq(False, False, False, False))
print("seen + near                    ", seq(True, False, False, False))
print("seen + near + f_aggr GEMM      ", seq(True, True, False, False))
print("... + rotating graphs          ", seq(True, True, True, False))
print("... + fresh out buffer         ", seq(True, True, True, True))
