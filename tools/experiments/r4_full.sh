#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
echo "== gpu tests"; timeout 1800 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "== smoke"; timeout 300 python3 __graft_entry__.py smoke 2>&1 | tail -2
echo "== hash"; timeout 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
run() { tag=$1; shift; timeout -s KILL 400 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
A="--steps 16 --warmup 3 --no-cpu-baseline --no-planner --no-config1"
run i3_1 python3 bench.py $A
run i2_1 python3 bench.py $A --inflight 2
run i3_2 python3 bench.py $A
run i2_2 python3 bench.py $A --inflight 2
