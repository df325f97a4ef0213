#!/bin/bash
# A/B on one box: pooled shares of the two long-candidate classes, batches in flight, chaining
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/ab_$tag.json || tail -3 gpurun_out/ab_$tag.err; }
A="--steps 10 --warmup 2 --no-cpu-baseline --no-config1"
run base1 python3 bench.py $A
run pool1 env TOPAY_BIG_POOL=1 python3 bench.py $A
run base2 python3 bench.py $A
run pool2 env TOPAY_BIG_POOL=1 python3 bench.py $A
run if3 python3 bench.py $A --inflight 3 --no-serial
run pool_if3 env TOPAY_BIG_POOL=1 python3 bench.py $A --inflight 3 --no-serial
run chain python3 bench.py $A --chain --no-serial
run pool_chain env TOPAY_BIG_POOL=1 python3 bench.py $A --chain --no-serial
