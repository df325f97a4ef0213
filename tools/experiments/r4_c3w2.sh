#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
X=$PWD/tools/libs/libtopay_exp.so
A="--steps 16 --warmup 3 --no-cpu-baseline --no-planner --no-config1"
run() { tag=$1; shift; timeout -s KILL 400 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
for rep in 1 2; do
run base_$rep env TOPAY_LIB=$X python3 bench.py $A
run c3w2_$rep env TOPAY_LIB=$X TOPAY_MW_C3=2 python3 bench.py $A
run c23w2_$rep env TOPAY_LIB=$X TOPAY_MW_C3=2 TOPAY_MW_C2=2 python3 bench.py $A
done
