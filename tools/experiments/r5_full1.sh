#!/bin/bash
# round 5: whole GPU suite on the tree (writes gpurun_out/agreement_stats_measured.json), exposed ESDF-gather latency on the 4 GB
# field (stamps build) at one and two waves per SIMD, product vs experiments build interleaved
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5f1; mkdir -p $O
timeout -s KILL 1800 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for S in 128 512; do echo "== hires stamps S=$S"; TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 900 python3 tools/gpu_stamps_hires.py $S 2>&1 | tail -3; done
echo "== cached maps, same build"; TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 600 python3 tools/gpu_stamps.py 512 2>&1 | tail -4
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2; do for v in prod exp; do
  if [ $v = exp ]; then export TOPAY_LIB=tools/libs/libtopay_exp.so; else unset TOPAY_LIB; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
