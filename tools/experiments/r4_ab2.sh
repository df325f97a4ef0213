#!/bin/bash
# the four-wave classes alone (N >= 33) and the serial step: last commit (prev) against the working tree (exp)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4u; mkdir -p $O
for v in prev exp prev exp; do
  echo "== $v"; TOPAY_LIB=tools/libs/libtopay_$v.so timeout 600 python3 tools/gpu_occupancy.py 1024 0 33 2>&1 | grep -E "^B |N 33-64"
done
for v in prev exp; do
  TOPAY_LIB=tools/libs/libtopay_$v.so timeout -s KILL 600 python3 bench.py --inflight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-planner --no-config1 > $O/s$v.json 2> $O/s$v.err; python3 tools/pj.py serial-$v < $O/s$v.json
done
