#!/bin/bash
# one-wave solver without the drain + barrier ahead of an evaluation (ss), the deeper prefetch ring (pf32), both (sspf32), against the
# shipped settings (exp); -DTOPAY_EXPERIMENTS builds, same bits, three interleaved rounds
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5ssync; mkdir -p $O
for v in ss sspf32; do echo "hash $v: $(TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so timeout 300 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"; done
for r in 1 2 3; do for v in exp ss pf32 sspf32; do
  TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so timeout -s KILL 300 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-planner > $O/$v$r.json 2> $O/$v$r.err
  python3 -c "
import json; j=json.load(open('$O/$v$r.json')); r=j['roofline']; c=j['config']['config1_latency']; print('$v round $r: %6.0f traj/s  %.1f ms/step  slot-seconds %.0f  serial %.0f  configs[1] %.1f / %.1f' % (j['value'], j['ms_per_step'], r['slot_seconds_per_step'], r['serial_steps']['ms_per_step'], c['solve_ms'], c['solve_ms_latency_mode']))" 2>/dev/null || tail -2 $O/$v$r.err
done; done
