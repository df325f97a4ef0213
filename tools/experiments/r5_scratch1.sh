#!/bin/bash
# round 5, manipulator block without stack traffic in its interface (pose inside, torques through the LDS pass buffer):
# result hash (must stay c4180d2ab9d1aef9), the evaluation's GPU tests, interleaved bench r4 vs tree, ESDF-gather kernel traffic
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5s1; mkdir -p $O
echo "hash tree: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (r4: c4180d2ab9d1aef9)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py tests/test_collision.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2; do for v in r4 tree; do
  if [ $v = r4 ]; then export TOPAY_LIB=tools/libs/libtopay_r4.so; else unset TOPAY_LIB; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
unset TOPAY_LIB
