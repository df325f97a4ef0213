#!/bin/bash
# compact LDS layout for the four-wave classes as well (65 / 81 KB instead of 83 / 108 KB at N = 42 / 63)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4n; mkdir -p $O
echo "hash exp:          $(TOPAY_LIB=tools/libs/libtopay_exp.so timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
echo "hash exp compact:  $(TOPAY_COMPACT_ALL=1 TOPAY_LIB=tools/libs/libtopay_exp.so timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
for r in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export TOPAY_COMPACT_ALL=1; else unset TOPAY_COMPACT_ALL; fi
  TOPAY_LIB=tools/libs/libtopay_exp.so timeout -s KILL 600 python3 bench.py --no-cpu-baseline --no-planner --no-config1 > $O/c$v$r.json 2> $O/c$v$r.err; python3 tools/pj.py compact$v-$r < $O/c$v$r.json
done; done
