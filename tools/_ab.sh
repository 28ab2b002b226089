cd /root/repo
timeout 300 python -m pytest tests -q -x -m gpu -k "matrix_cores or backward_is_deterministic or full_size_backward or golden" 2>&1 | tail -3
for r in 1 2; do for l in libuavgnn_prev.so libuavgnn.so; do
echo $l; UAVGNN_LIB=$PWD/uav_bs_ctrl_amd/csrc/$l python tools/kbench.py --dists dense --reps 3 2>&1 | grep "bwd dense seen"
done; done
REPS=50 REL=1 CASES="[(4096,64,64),(8192,16,130)]" python tools/k1_bwd_repro.py 2>&1 | grep -v "dattn\|dW_r\|db_r" | tail -12
