#!/bin/bash
# Round 3: share of the smallest class (its workgroups cannot take over other queues) + config1 in latency mode.
mkdir -p gpurun_out/r3o
run() { tag=$1; shift; timeout -s KILL 500 "$@" > gpurun_out/r3o/b_$tag.json 2> gpurun_out/r3o/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3o/b_$tag.json || tail -3 gpurun_out/r3o/b_$tag.err; }
A="--no-cpu-baseline --no-planner"
run b100 python3 bench.py $A
run b090 env TOPAY_SHARE_BIAS0=0.9 python3 bench.py $A --no-config1
run b080 env TOPAY_SHARE_BIAS0=0.8 python3 bench.py $A --no-config1
run b070 env TOPAY_SHARE_BIAS0=0.7 python3 bench.py $A --no-config1
run b100b python3 bench.py $A --no-config1
python3 - <<'P'
import json
j=json.loads(open("gpurun_out/r3o/b_b100.json").read().strip().split("\n")[-1]); print(j["config"]["config1_latency"])
P
