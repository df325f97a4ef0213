#!/bin/bash
# batches in flight per GPU, now that run-to-run noise is gone: 2 / 3 (default) / 4 / 5, three interleaved rounds of 12 steps
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5inflight; mkdir -p $O
for r in 1 2 3; do for n in 2 3 4 5; do
  timeout -s KILL 300 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-planner --no-config1 --no-serial --inflight $n > $O/i$n$r.json 2> $O/i$n$r.err
  python3 -c "
import json; j=json.load(open('$O/i$n$r.json')); r=j['roofline']; print('inflight $n round $r: %6.0f traj/s  %.1f ms/step  slot-seconds %.0f' % (j['value'], j['ms_per_step'], r['slot_seconds_per_step']))" 2>/dev/null || tail -2 $O/i$n$r.err
done; done
