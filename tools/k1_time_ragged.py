import os, sys, torch as th
sys.path.insert(0, "/root/repo")
from bench import synth_batch_gpu
from uav_bs_ctrl_amd import _lib as L
from uav_bs_ctrl_amd.agents.gnn_agents import GATv2Conv
dev = th.device("cuda"); gen = th.Generator(device=dev); gen.manual_seed(0); th.manual_seed(0)
lib, st = L.lib(), L.stream()
ps = []
for FS in (4, 2):
    c = GATv2Conv((FS, 2), 64, 4).to(dev)
    ps.append([t.detach().contiguous() for t in (c.fc_src.weight, c.fc_src.bias, c.fc_dst.weight, c.fc_dst.bias, c.attn, c.res_fc.weight, c.res_fc.bias)])
for dist, B in (("nz", 4096), ("dense", 4096), ("nz", 4096 * 51)):
    hb = synth_batch_gpu(B, 8, 80, dist, dev, gen)
    xs, so = hb.relation_segments("seen"); xn, no = hb.relation_segments("near"); order = hb.relation_order("seen"); x_a = hb.agent_feat(); N = x_a.shape[0]
    out = th.empty(N, 512, device=dev)
    a_s, a_n = th.empty(max(xs.shape[0], 1), 4, device=dev), th.empty(max(xn.shape[0], 1), 4, device=dev)
    for save in (False, True):
        for ph in (3, 3 | 256, 1, 2):
            def run():
                rc = lib.uavgnn_gatv2_hetero_fwd_phases(xs.data_ptr(), xs.shape[0], so.data_ptr(), L.ptr(order), xn.data_ptr(), xn.shape[0], no.data_ptr(), x_a.data_ptr(), N, L.ptr_array(ps[0]), L.ptr_array(ps[1]), 4, 64, 0.2, out.data_ptr(), 512, a_s.data_ptr() if save else None, a_n.data_ptr() if save else None, ph, st)
                assert rc == 0
            for _ in range(3): run()
            th.cuda.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            reps = 20 if B == 4096 else 3
            e0.record()
            for _ in range(reps): run()
            e1.record(); th.cuda.synchronize()
            print(f"{dist:6s} B={B:7d} save={int(save)} phases={ph:4d}: {e0.elapsed_time(e1) / reps * 1e3:9.1f} us  (E_seen {xs.shape[0]})")
