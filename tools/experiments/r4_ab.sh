#!/bin/bash
# interleaved bench: tools/libs/libtopay_prev.so (the last commit) against libtopay_exp.so (the working tree)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4s; mkdir -p $O
for r in 1 2; do for v in prev exp; do
  TOPAY_LIB=tools/libs/libtopay_$v.so timeout -s KILL 600 python3 bench.py --no-cpu-baseline --no-planner --no-config1 > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json
done; done
