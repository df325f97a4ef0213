#!/bin/bash
# pipelined residency: timeline of three batches in flight; oversubscription of the launch grids again (experiments build)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5pipe; mkdir -p $O
timeout 600 python3 tools/gpu_pipeline.py 1024 3 9 2>&1 | tail -40
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
export TOPAY_LIB=tools/libs/libtopay_exp.so
for r in 1 2; do for ov in 1.08 1.4 1.8 2.4; do
  TOPAY_OVERSUBSCRIBE=$ov timeout -s KILL 400 python3 bench.py $A > $O/ov${ov}_$r.json 2> $O/ov${ov}_$r.err; python3 tools/pj.py ov${ov}_$r < $O/ov${ov}_$r.json || tail -3 $O/ov${ov}_$r.err
done; done
