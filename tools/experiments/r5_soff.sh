#!/bin/bash
# history rows of the two-loop recursion through one descriptor per block and the instruction's scalar offset: hash (must stay
# 0dbe2e2a1efb9921: the range check must still zero the pairs beyond a row), solver tests, interleaved bench against the build before
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5soff; mkdir -p $O
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (before: 0dbe2e2a1efb9921)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2 3; do for v in prev9 tree; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
unset TOPAY_LIB
