#!/bin/bash
# Round 3: A/B of the compact LDS layout (register-accumulator evaluation) for the common one-wave classes.
mkdir -p gpurun_out/r3g
echo "== hash, one wave everywhere, compact layout for N <= 32 (round 2: 0dbe2e2a1efb9921)"
TOPAY_COMPACT=1 TOPAY_MW_C4=1 TOPAY_MW_C5=1 timeout -s KILL 300 python3 tools/gpu_hashrun.py 2>&1 | tail -1
run() { tag=$1; shift; timeout -s KILL 400 "$@" > gpurun_out/r3g/b_$tag.json 2> gpurun_out/r3g/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3g/b_$tag.json || tail -3 gpurun_out/r3g/b_$tag.err; }
A="--steps 16 --warmup 3 --no-cpu-baseline --no-config1"
run base python3 bench.py $A
run compact env TOPAY_COMPACT=1 python3 bench.py $A
run base2 python3 bench.py $A
run compact2 env TOPAY_COMPACT=1 python3 bench.py $A
TOPAY_COMPACT=1 timeout -s KILL 300 python3 tools/gpu_pipeline.py 1024 3 10 2>&1 | grep "steps depth\|max resident\|total resident\|idle slot"
