#!/bin/bash
# Round 4, first GPU call: result hash of the two-waves-per-SIMD build against the round-3 library, parity tests, A/B bench.
O=gpurun_out/r4a; mkdir -p $O
R3=$PWD/tools/libs/libtopay_r3.so
echo "== hash new"; timeout 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
echo "== hash r3";  TOPAY_LIB=$R3 timeout 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
echo "== parity tests"; timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_multiwave.py -x -q -m gpu 2>&1 | tail -5
run() { tag=$1; shift; timeout -s KILL 400 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
A="--steps 12 --warmup 3 --no-cpu-baseline --no-planner --no-config1"
run r3_1 env TOPAY_LIB=$R3 python3 bench.py $A
run new_1 python3 bench.py $A
run r3_2 env TOPAY_LIB=$R3 python3 bench.py $A
run new_2 python3 bench.py $A
