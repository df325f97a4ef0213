#!/bin/bash
# last run of the round on the committed sources: result hash, determinism, the default bench line (with the traffic of the committed
# profile quoted) and the line through the RCCL code path
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (expected: 0dbe2e2a1efb9921)"
S=512 timeout 600 python3 tools/gpu_determinism.py 2>&1 | tail -1
timeout -s KILL 600 python3 bench.py > gpurun_out/final_default.json 2> gpurun_out/final_default.err; python3 tools/pj.py default < gpurun_out/final_default.json
timeout -s KILL 400 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-config1 --no-serial > gpurun_out/final_dist.json 2> gpurun_out/final_dist.err
grep '^{' gpurun_out/final_dist.json | tail -1 | python3 tools/pj.py dist_default_env
