#!/bin/bash
# tip-to-base manipulator block against the stored-forces version of the previous commit
O=gpurun_out/r4g; mkdir -p $O
for l in prev_stamps stamps; do echo "== $l S=512"; TOPAY_LIB=$PWD/tools/libs/libtopay_$l.so timeout 300 python3 tools/gpu_stamps.py 512 2>&1 | grep "kernel\|mani\|lbfgs\|twoloop\|total"; done
echo "== hash new"; timeout 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
A="--steps 12 --warmup 3 --no-cpu-baseline --no-planner --no-config1"
run() { tag=$1; shift; timeout -s KILL 400 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
run prev_1 env TOPAY_LIB=$PWD/tools/libs/libtopay_prev.so python3 bench.py $A
run new_1 python3 bench.py $A
run prev_2 env TOPAY_LIB=$PWD/tools/libs/libtopay_prev.so python3 bench.py $A
run new_2 python3 bench.py $A
echo "== parity tests"; timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_multiwave.py -x -q -m gpu 2>&1 | tail -3
