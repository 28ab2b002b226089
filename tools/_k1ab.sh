cd /root/repo
for d in dense env ragged; do timeout 300 python tools/k1_check.py --dist $d --B 256 --save | grep -v "run . vs"; done
timeout 100 python tools/k1_check.py --dist env --B 8192 --n 4 | grep -v "run . vs"
timeout 100 python tools/k1_check.py --dist env --B 4096 --n 8 --M 80 --no-order | grep -v "run . vs"
for rep in 1 2 3; do
for v in head 10 q2 q3 q4; do
  if [ $v != head ]; then echo "== $v image rep $rep"; K1_IMAGE=1 timeout 120 tools/ubench/bin/k1v_$v env 4096 50 | grep -E "phases +(0|1|2|3):"; else
  echo "== $v plain rep $rep"; timeout 120 tools/ubench/bin/k1v_$v env 4096 50 | grep -E "phases +(0|1|2|3):"; fi
done
done
for v in head 10 q2 q3 q4; do echo "== $v big"; K1_IMAGE=1 timeout 120 tools/ubench/bin/k1v_$v env 208896 5 | grep -E "phases +(2|3):"; echo "== $v dense"; K1_IMAGE=1 timeout 120 tools/ubench/bin/k1v_$v dense 4096 50 | grep -E "phases +(1|2|3):"; done
K1_IMAGE=1 timeout 120 tools/ubench/bin/k1v_q3 env 4096 50 | grep -A9 "time line"
