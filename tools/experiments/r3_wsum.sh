#!/bin/bash
# Round 3: wave_sum with DPP row broadcasts across the rows (20 instead of 23 instructions): result hash, stamps, bench A/B.
mkdir -p gpurun_out/r3u
timeout -s KILL 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
for v in base new; do echo == $v; TOPAY_LIB=$PWD/tools/libs/libtopay_stamps_$v.so timeout -s KILL 300 python3 tools/gpu_stamps.py 2>&1 | grep -E "lbfgs|twoloop|total"; done
run() { tag=$1; shift; timeout -s KILL 500 "$@" > gpurun_out/r3u/b_$tag.json 2> gpurun_out/r3u/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3u/b_$tag.json || tail -3 gpurun_out/r3u/b_$tag.err; }
A="--steps 16 --warmup 3 --no-cpu-baseline --no-planner --no-config1"
B=$PWD/tools/libs/libtopay_base.so
run base1 env TOPAY_LIB=$B python3 bench.py $A
run new1 python3 bench.py $A
run base2 env TOPAY_LIB=$B python3 bench.py $A
run new2 python3 bench.py $A
