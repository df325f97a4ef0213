#!/bin/bash
# round 5: sphere-split manipulator block for short passes: the evaluation's parity tests, interleaved bench against the same
# sources built with -DTOPAY_NO_SPLIT_PASS (experiments build otherwise), per-evaluation time by N (eval hook) both ways
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5sp1; mkdir -p $O
echo "hash tree: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py tests/test_collision.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2 3; do for v in nosplit split; do
  if [ $v = nosplit ]; then export TOPAY_LIB=tools/libs/libtopay_nosplit.so; else unset TOPAY_LIB; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
unset TOPAY_LIB
