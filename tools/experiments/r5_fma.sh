#!/bin/bash
# product-sums of the evaluation written as fma (sin / cos kernels, ESDF fractions, matrix products, penalty accumulations):
# new hash, the whole GPU suite, interleaved bench against the unfused build (tools/libs/libtopay_nofma.so)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5fma; mkdir -p $O
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (unfused: 0dbe2e2a1efb9921)"
timeout -s KILL 1800 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2 3; do for v in nofma fma; do
  if [ $v = fma ]; then unset TOPAY_LIB; else export TOPAY_LIB=tools/libs/libtopay_nofma.so; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
unset TOPAY_LIB
