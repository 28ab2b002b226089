import sys, hashlib
sys.path.insert(0, "/root/repo")
import torch as th, types
import bench
from uav_bs_ctrl_amd.agents.gnn_agents import GraphObservationEncoder
dev = th.device("cuda")
th.manual_seed(0)
gen = th.Generator(device=dev); gen.manual_seed(1)
for dist, B in (("dense", 512), ("env", 2048)):
    g = bench.synth_batch_gpu(B, 8, 80, dist, dev, gen)
    enc = GraphObservationEncoder(dict(agent=2, ubs=2, gt=4), types.SimpleNamespace(hidden_size=256, n_heads=4, n_layers=2)).to(dev)
    y = enc(g)
    (y * th.randn_like(y)).sum().backward()
    hsh = hashlib.sha256()
    for p in enc.parameters():
        hsh.update(p.grad.detach().cpu().numpy().tobytes())
    print(dist, hsh.hexdigest()[:16])
