#!/bin/bash
# pipelining knobs under the two-waves build: batches in flight x oversubscription x assumed gain
O=gpurun_out/r4d; mkdir -p $O
X=$PWD/tools/libs/libtopay_exp.so
A="--steps 10 --warmup 3 --no-cpu-baseline --no-planner --no-config1 --no-serial"
run() { tag=$1; shift; timeout -s KILL 300 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 -c "
import json,sys
j=json.load(open('$O/b_$tag.json')); print('$tag', round(j['value']), round(j['ms_per_step'],1), 'util', round(j['roofline']['slot_utilisation'],3))" || tail -2 $O/b_$tag.err; }
for inf in 2 3 4; do for ov in 1.0 1.08 1.3; do
  run i${inf}_o${ov} env TOPAY_LIB=$X TOPAY_OVERSUBSCRIBE=$ov python3 bench.py $A --inflight $inf
done; done
for g in 1.0 1.2 1.7 2.0; do run g$g env TOPAY_LIB=$X TOPAY_OCC2_GAIN=$g python3 bench.py $A; done
run gate0 env TOPAY_LIB=$X TOPAY_DISPATCH_GATE=0 python3 bench.py $A
run bias1 env TOPAY_LIB=$X TOPAY_SHARE_BIAS0=1.0 python3 bench.py $A
