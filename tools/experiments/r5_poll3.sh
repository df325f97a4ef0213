#!/bin/bash
# how much do the remaining bus polls cost?  The shipped library (every 8th evaluation) against builds that poll every 32nd and never
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5poll3; mkdir -p $O
A="--steps 10 --warmup 2 --no-cpu-baseline --no-planner --no-config1"
for r in 1 2 3; do for v in tree poll31 poll0; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  timeout -s KILL 300 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err
  python3 -c "
import json; j=json.load(open('$O/$v$r.json')); r=j['roofline']; print('$v$r %6.0f traj/s  %.1f ms/step  slot-seconds %.0f  serial %.0f ms' % (j['value'], j['ms_per_step'], r['slot_seconds_per_step'], r['serial_steps']['ms_per_step']))"
done; done
