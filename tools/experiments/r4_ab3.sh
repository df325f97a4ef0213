#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
A="--steps 10 --warmup 3 --no-cpu-baseline --no-planner --no-config1 --no-serial --inflight 2"
run() { tag=$1; shift; timeout -s KILL 300 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 -c "
import json,sys
j=json.load(open('$O/b_$tag.json')); print('$tag', round(j['value']), round(j['ms_per_step'],1))" || tail -2 $O/b_$tag.err; }
for rep in 1 2; do for v in exp la2 nofence; do
  run ${v}_$rep env TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so TOPAY_OCC2_GAIN=1.7 python3 bench.py $A
done; done
