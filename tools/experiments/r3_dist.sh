#!/bin/bash
# Round 3: the RCCL code path on one GPU against the plain default (streams per context 7 -> 6).
mkdir -p gpurun_out/r3r
export HSA_ENABLE_IPC_MODE_LEGACY=0
A="--gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-config1 --no-serial --no-planner"
for i in 1 2; do
timeout -s KILL 400 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py $A > gpurun_out/r3r/dist$i.json 2> gpurun_out/r3r/dist$i.err
grep '^{' gpurun_out/r3r/dist$i.json | tail -1 | python3 tools/pj.py dist$i
timeout -s KILL 400 python3 bench.py $A > gpurun_out/r3r/plain$i.json 2> gpurun_out/r3r/plain$i.err
python3 tools/pj.py plain$i < gpurun_out/r3r/plain$i.json
done
timeout -s KILL 600 python -m pytest tests/test_multiwave.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
