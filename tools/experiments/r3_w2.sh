#!/bin/bash
# Round 3: a build limited to 256 registers (TOPAY_WAVES_PER_EU=2, offload LTO) with one and with two waves per SIMD.
mkdir -p gpurun_out/r3s
run() { tag=$1; shift; timeout -s KILL 500 "$@" > gpurun_out/r3s/b_$tag.json 2> gpurun_out/r3s/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3s/b_$tag.json || tail -3 gpurun_out/r3s/b_$tag.err; }
A="--steps 12 --warmup 2 --no-cpu-baseline --no-planner --no-config1"
run base python3 bench.py $A
run w2_slots1 env TOPAY_LIB=$PWD/tools/libs/libtopay_w2.so python3 bench.py $A
run w2_slots2 env TOPAY_LIB=$PWD/tools/libs/libtopay_w2.so TOPAY_SLOTS_PER_SIMD=2 python3 bench.py $A
