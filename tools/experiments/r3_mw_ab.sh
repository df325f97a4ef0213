#!/bin/bash
# Round 3: GPU checks of the several-waves-per-trajectory kernels + A/B of the waves per trajectory of the long classes.
mkdir -p gpurun_out/r3c
echo "== hash, every class on one wave (round 2: 0dbe2e2a1efb9921): O3 / O2 / O1"
for l in "" O2 O1; do TOPAY_LIB=${l:+$PWD/tools/libs/libtopay_$l.so} TOPAY_MW_C4=1 TOPAY_MW_C5=1 timeout -s KILL 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1; done
echo "== hash default (long classes on four waves): O3 / O2 / O1"
for l in "" O2 O1; do TOPAY_LIB=${l:+$PWD/tools/libs/libtopay_$l.so} timeout -s KILL 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1; done
timeout -s KILL 1500 python -m pytest tests/test_multiwave.py tests/test_outputs_gpu.py -m gpu -x -q > gpurun_out/r3c/t_mw.log 2>&1; tail -3 gpurun_out/r3c/t_mw.log
run() { tag=$1; shift; timeout -s KILL 400 "$@" > gpurun_out/r3c/b_$tag.json 2> gpurun_out/r3c/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3c/b_$tag.json || tail -3 gpurun_out/r3c/b_$tag.err; }
A="--steps 16 --warmup 3 --no-cpu-baseline"
run w11 env TOPAY_MW_C4=1 TOPAY_MW_C5=1 python3 bench.py $A
run w44 python3 bench.py $A
run w24 env TOPAY_MW_C4=2 TOPAY_MW_C5=4 python3 bench.py $A
run w14 env TOPAY_MW_C4=1 TOPAY_MW_C5=4 python3 bench.py $A
run w22 env TOPAY_MW_C4=2 TOPAY_MW_C5=2 python3 bench.py $A
run w11b env TOPAY_MW_C4=1 TOPAY_MW_C5=1 python3 bench.py $A
run w44b python3 bench.py $A
