#!/bin/bash
# The multi-GPU code path (RCCL record gather on its own stream beside the solves) on one GPU, against the plain path,
# with 16 and 24 hardware queues: does the gather's stream collide with a solve launch's queue?
run() { tag=$1; shift; timeout -s KILL 400 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; grep '^{' gpurun_out/ab_$tag.json | tail -1 | python3 tools/pj.py "$tag" || tail -3 gpurun_out/ab_$tag.err; }
A="--gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-serial"
T="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run plain16 python3 bench.py $A
run dist16 env TOPAY_FORCE_DIST=1 $T bench.py $A
run dist24 env TOPAY_FORCE_DIST=1 GPU_MAX_HW_QUEUES=24 $T bench.py $A
run plain24 env GPU_MAX_HW_QUEUES=24 python3 bench.py $A
run dist16b env TOPAY_FORCE_DIST=1 $T bench.py $A
run dist24b env TOPAY_FORCE_DIST=1 GPU_MAX_HW_QUEUES=24 $T bench.py $A
run dist20 env TOPAY_FORCE_DIST=1 GPU_MAX_HW_QUEUES=20 $T bench.py $A
