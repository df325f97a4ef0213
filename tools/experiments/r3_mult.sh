#!/bin/bash
mkdir -p gpurun_out/r3p
run() { tag=$1; shift; timeout -s KILL 500 "$@" > gpurun_out/r3p/b_$tag.json 2> gpurun_out/r3p/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3p/b_$tag.json || tail -3 gpurun_out/r3p/b_$tag.err; }
A="--no-cpu-baseline --no-planner --no-config1"
run base python3 bench.py $A
run m1 env TOPAY_SHARE_MULT=1,1.09,1.04,1.07,0.9,0.93,1 python3 bench.py $A
run m2 env TOPAY_SHARE_MULT=1,1.15,1.08,1.12,0.85,0.9,1 python3 bench.py $A
run m3 env TOPAY_SHARE_BIAS0=0.7 TOPAY_SHARE_MULT=1,1.2,1.1,1.15,0.8,0.85,1 python3 bench.py $A
run base2 python3 bench.py $A
