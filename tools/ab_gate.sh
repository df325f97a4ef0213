#!/bin/bash
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/ab_$tag.json || tail -3 gpurun_out/ab_$tag.err; }
A="--steps 16 --warmup 3 --no-cpu-baseline --no-config1 --no-serial"
run gate_if3 python3 bench.py $A
run nogate_if3 env TOPAY_DISPATCH_GATE=0 python3 bench.py $A
run nogate_if4 env TOPAY_DISPATCH_GATE=0 python3 bench.py $A --inflight 4
run gate_if4 python3 bench.py $A --inflight 4
run nogate_if2 env TOPAY_DISPATCH_GATE=0 python3 bench.py $A --inflight 2
run gate_if3b python3 bench.py $A
