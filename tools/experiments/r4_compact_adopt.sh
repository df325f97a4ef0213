#!/bin/bash
# the adjoint solve in the coefficients' block for every class (the separate LDS layout removed): hash, the tests that touch the
# several-waves evaluation, interleaved bench against the previous build, residency and time per N bin
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4o; mkdir -p $O
echo "hash product: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for r in 1 2; do for v in prev exp; do
  TOPAY_LIB=tools/libs/libtopay_$v.so timeout -s KILL 600 python3 bench.py --no-cpu-baseline --no-planner --no-config1 > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json
done; done
echo "== occupancy"; timeout 600 python3 tools/gpu_occupancy.py 1024 2>&1 | tail -14
