#!/bin/bash
# pipelining knobs again after the LDS relief (1.7 resident waves per SIMD): batches in flight, oversubscription, assumed gain, gate
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4q; mkdir -p $O
X=$PWD/tools/libs/libtopay_exp.so
A="--steps 15 --warmup 3 --no-cpu-baseline --no-planner --no-config1 --no-serial"
run() { tag=$1; shift; timeout -s KILL 300 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 -c "
import json,sys
j=json.load(open('$O/b_$tag.json')); print('$tag', round(j['value']), round(j['ms_per_step'],1), 'util', round(j['roofline']['slot_utilisation'],3))" || tail -2 $O/b_$tag.err; }
run base1 env TOPAY_LIB=$X python3 bench.py $A
for inf in 2 4; do run i${inf} env TOPAY_LIB=$X python3 bench.py $A --inflight $inf; done
for ov in 1.0 1.2 1.4; do run o$ov env TOPAY_LIB=$X TOPAY_OVERSUBSCRIBE=$ov python3 bench.py $A; done
run base2 env TOPAY_LIB=$X python3 bench.py $A
for g in 1.3 1.5 2.0; do run g$g env TOPAY_LIB=$X TOPAY_OCC2_GAIN=$g python3 bench.py $A; done
run gate0 env TOPAY_LIB=$X TOPAY_DISPATCH_GATE=0 python3 bench.py $A
run bias1 env TOPAY_LIB=$X TOPAY_SHARE_BIAS0=1.0 python3 bench.py $A
run base3 env TOPAY_LIB=$X python3 bench.py $A
