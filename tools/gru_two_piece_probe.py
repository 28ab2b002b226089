import sys; sys.path.insert(0, "/root/repo")
import torch as th
from uav_bs_ctrl_amd import ops, enable_tuned_gemms
enable_tuned_gemms()
N,H,M=32768,256,64
cell=th.nn.GRUCell(H+M,H).cuda()
x,c,h=th.randn(N,H,device="cuda"),th.randn(N,M,device="cuda"),th.randn(N,H,device="cuda")
cat=th.cat((x,c),1)
def t(fn,reps=50):
    for _ in range(3): fn()
    th.cuda.synchronize(); e0,e1=th.cuda.Event(enable_timing=True),th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); th.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
with th.no_grad():
    for r in range(2):
        print("one piece", round(t(lambda: ops._gru_cell_launch(cat,h,cell.weight_ih,cell.bias_ih,cell.weight_hh,cell.bias_hh,save=False)),1),
              "two pieces", round(t(lambda: ops._gru_cell_launch(x,h,cell.weight_ih,cell.bias_ih,cell.weight_hh,cell.bias_hh,save=False,inp2=c)),1))
