"""Launches the round-5 streaming kernels a few times at C3 size (for tools/pmc.sh): the Q head, the gate-gradient kernel with column
sums, the ReLU backward fused with the bias gradient.  Algorithmic bytes are printed for comparison with FETCH_SIZE / WRITE_SIZE."""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uav_bs_ctrl_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = th.device("cuda")
N, H, A = 32768, 256, 9
h = th.randn(N, H, device=dev)
W = th.randn(A, H, device=dev) * 0.1
b = th.zeros(A, device=dev)
q = th.empty(N, A, device=dev)
pre = th.randn(N, 4 * H, device=dev)
dho = th.randn(N, H, device=dev)
dq = th.randn(N, A, device=dev)
G = lib.uavgnn_gru_gates_bwd_sum_rows(N, H)
d_gi, d_gh, d_h = th.empty(N, 3 * H, device=dev), th.empty(N, 3 * H, device=dev), th.empty(N, H, device=dev)
sums = th.empty(G, 4 * H, device=dev)
n2, C = 51 * N, 256
dy, y, out = th.randn(n2, C, device=dev), th.relu(th.randn(n2, C, device=dev)), th.empty(n2, C, device=dev)
acc = th.zeros(256, C, device=dev)
for _ in range(5):
    L.check(lib.uavgnn_head_fwd(h.data_ptr(), H, N, H, W.data_ptr(), H, b.data_ptr(), A, q.data_ptr(), A, L.stream()), "head")
    L.check(lib.uavgnn_gru_gates_bwd_fused_sums(pre.data_ptr(), h.data_ptr(), dho.data_ptr(), dq.data_ptr(), A, W.data_ptr(), N, H,
                                                d_gi.data_ptr(), d_gh.data_ptr(), d_h.data_ptr(), sums.data_ptr(), L.stream()), "gates")
    L.check(lib.uavgnn_relu_bwd_colsum(dy.data_ptr(), C, y.data_ptr(), C, out.data_ptr(), C, n2, C, acc.data_ptr(), 256, L.stream()), "relu")
th.cuda.synchronize()
print(f"# algorithmic bytes per launch: head_fwd read {4 * N * H / 1e6:.1f} MB, write {4 * N * A / 1e6:.2f} MB | gates bwd (+ sums) read "
      f"{4 * N * (4 * H + 2 * H + A) / 1e6:.1f} MB, write {4 * (N * 7 * H + G * 4 * H) / 1e6:.1f} MB | relu_bwd_colsum read {8 * n2 * C / 1e6:.0f} MB, "
      f"write {4 * n2 * C / 1e6:.0f} MB   (FETCH_SIZE / WRITE_SIZE: see MI355X_MICROARCH.md for the unit)")
