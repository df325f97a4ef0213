#!/bin/bash
# split pass, second look: no split / split with scheduling fences (G = 4 and 2) / G = 4 only; serial and pipelined; evaluation time by N
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5sp2; mkdir -p $O
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2; do for v in nosplit split g4only; do
  if [ $v = split ]; then unset TOPAY_LIB; else export TOPAY_LIB=tools/libs/libtopay_$v.so; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
for v in nosplit split; do
  if [ $v = split ]; then unset TOPAY_LIB; else export TOPAY_LIB=tools/libs/libtopay_$v.so; fi
  echo "== evaluation by N, $v"; timeout 600 python3 tools/gpu_eval_by_n.py 2>&1 | tail -15
done
