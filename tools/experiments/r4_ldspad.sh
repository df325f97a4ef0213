#!/bin/bash
# what one workgroup less per compute unit costs: classes N <= 21 (6 -> 5 per CU) and N <= 32 (4 -> 3) padded
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4p; mkdir -p $O
for r in 1 2; do for v in none 21:700 32:1900 21:700,32:1900; do
  if [ $v = none ]; then unset TOPAY_LDS_PAD; else export TOPAY_LDS_PAD=$v; fi
  TOPAY_LIB=tools/libs/libtopay_exp.so timeout -s KILL 600 python3 bench.py --no-cpu-baseline --no-planner --no-config1 > $O/p$v$r.json 2> $O/p$v$r.err; python3 tools/pj.py pad$v-$r < $O/p$v$r.json
done; done
