cd /root/repo
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16x2 or learner_update_at_exp3" 2>&1 | tail -40 > gpurun_out/r06_tests_c.txt
cat gpurun_out/r06_tests_c.txt
python tools/h2_probe.py > gpurun_out/r06_h2_probe.txt 2>&1; cat gpurun_out/r06_h2_probe.txt
