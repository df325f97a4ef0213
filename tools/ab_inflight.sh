#!/bin/bash
# A/B on one box: batches in flight and hardware queues
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/ab_$tag.json || tail -3 gpurun_out/ab_$tag.err; }
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-serial"
run if2 python3 bench.py $A --inflight 2
run if3 python3 bench.py $A --inflight 3
run if4 python3 bench.py $A --inflight 4
run if5 python3 bench.py $A --inflight 5
run if3q24 env GPU_MAX_HW_QUEUES=24 python3 bench.py $A --inflight 3
run if4q24 env GPU_MAX_HW_QUEUES=24 python3 bench.py $A --inflight 4
run if3q32 env GPU_MAX_HW_QUEUES=32 python3 bench.py $A --inflight 3
run if3b python3 bench.py $A --inflight 3
