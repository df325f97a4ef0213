#!/bin/bash
# solver vectors carried in registers from the convergence test to the first trial point (two HBM round trips per iteration instead of
# seven, one per trial instead of two), no drain before an evaluation: hash twice (must stay 0dbe2e2a1efb9921), determinism, stamps,
# bench previous build (tools/libs/libtopay_prev13.so) against the tree, interleaved
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5carry; mkdir -p $O
for r in 1 2; do echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (before: 0dbe2e2a1efb9921)"; done
S=512 timeout 600 python3 tools/gpu_determinism.py 2>&1 | tail -3
TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 600 python3 tools/gpu_stamps_cfg1.py 2>&1 | grep -v "^     (" | tail -34
TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 600 python3 tools/gpu_stamps.py 512 2>&1 | grep -v "^   (" | tail -16
A="--steps 12 --warmup 3 --no-cpu-baseline --no-planner"
for r in 1 2 3; do for v in prev13 tree; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
  python3 -c "
import json,sys; j=json.load(open('$O/$v$r.json')); c=j['config']['config1_latency']; print('   configs[1] %.1f ms default, %.1f ms helper waves' % (c['solve_ms'], c['solve_ms_latency_mode']))"
done; done
unset TOPAY_LIB
