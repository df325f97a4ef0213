#!/bin/bash
mkdir -p gpurun_out/r3q
run() { tag=$1; shift; timeout -s KILL 500 "$@" > gpurun_out/r3q/b_$tag.json 2> gpurun_out/r3q/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3q/b_$tag.json || tail -3 gpurun_out/r3q/b_$tag.err; }
A="--no-cpu-baseline --no-planner --no-config1 --no-serial"
run early python3 bench.py $A
run late env TOPAY_EARLY_BIG=0 python3 bench.py $A
run early2 python3 bench.py $A
run late2 env TOPAY_EARLY_BIG=0 python3 bench.py $A
timeout -s KILL 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
