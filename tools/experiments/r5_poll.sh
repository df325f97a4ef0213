#!/bin/bash
# the host's cancel flag polled at every eighth evaluation instead of every one: hash (must stay 0dbe2e2a1efb9921), cancellation
# tests, solver sub-stamps, planning call, bench previous build (tools/libs/libtopay_prev14.so) against the tree, interleaved
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5poll; mkdir -p $O
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (before: 0dbe2e2a1efb9921)"
timeout -s KILL 900 python3 -m pytest tests/test_cancel.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -8
for S in 128 512; do echo "== S=$S"; TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 900 python3 tools/gpu_stamps.py $S 2>&1 | tail -24 | head -18 | grep -v "(-)"; done
A="--steps 8 --warmup 2 --no-cpu-baseline --no-planner"
for r in 1 2 3 4; do for v in prev14 tree; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  timeout -s KILL 300 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err
  python3 -c "
import json; j=json.load(open('$O/$v$r.json')); r=j['roofline']; c=j['config']['config1_latency']; print('$v$r %6.0f traj/s  %.1f ms/step  slot-seconds %.0f  serial %.0f ms  configs[1] %.1f / %.1f ms' % (j['value'], j['ms_per_step'], r['slot_seconds_per_step'], r['serial_steps']['ms_per_step'], c['solve_ms'], c['solve_ms_latency_mode']))"
done; done
