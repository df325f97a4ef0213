#!/bin/bash
# round 5: one-wave solver + four-wave evaluations for the long classes (k_long5 / k_long10) and the joints' cosines parked in LDS.
# hash (one-wave candidates unchanged; the long ones move), multi-wave + parity tests, A/B in the experiments build:
# TOPAY_LONG_SOLVER=4 (round 4's four-wave solver) against the default, interleaved; long classes alone (tools/gpu_bigN_time.py)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5l1; mkdir -p $O
echo "hash tree: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (r4: c4180d2ab9d1aef9)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
export TOPAY_LIB=tools/libs/libtopay_exp.so
for r in 1 2; do for v in w4 w1; do
  if [ $v = w4 ]; then export TOPAY_LONG_SOLVER=4; else unset TOPAY_LONG_SOLVER; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
unset TOPAY_LONG_SOLVER
echo "== long candidates alone, four-wave solver then one-wave solver"
TOPAY_LONG_SOLVER=4 timeout 300 python3 tools/gpu_bigN_time.py 36.0 50.0 66.0 100.0 133.0 2>&1 | tail -6
timeout 300 python3 tools/gpu_bigN_time.py 36.0 50.0 66.0 100.0 133.0 2>&1 | tail -6
unset TOPAY_LIB
timeout -s KILL 400 python3 bench.py $A > $O/prod.json 2> $O/prod.err; python3 tools/pj.py product < $O/prod.json
