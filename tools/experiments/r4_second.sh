#!/bin/bash
# lean one-wave LDS plan: hash, occupancy per class, A/B bench
O=gpurun_out/r4b; mkdir -p $O
R3=$PWD/tools/libs/libtopay_r3.so
echo "== hash new"; timeout 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
echo "== new, N<=10 only"; timeout 300 python3 tools/gpu_occupancy.py 1024 10
echo "== new, 11..15"; timeout 300 python3 tools/gpu_occupancy.py 1024 15 11
echo "== new, all"; timeout 300 python3 tools/gpu_occupancy.py 1024
run() { tag=$1; shift; timeout -s KILL 400 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
A="--steps 12 --warmup 3 --no-cpu-baseline --no-planner --no-config1"
run r3_1 env TOPAY_LIB=$R3 python3 bench.py $A
run new_1 python3 bench.py $A
run new_2 python3 bench.py $A
