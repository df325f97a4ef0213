#!/bin/bash
# round 5, final sources: the budget test again, the default bench lines on a fresh box, then the remaining profile pieces on these
# sources (SQ counters, the ESDF-gather kernel alone, configs[4] traffic and bench line)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 900 python3 -m pytest tests/test_cancel.py -m gpu -q 2>&1 | tail -2
timeout -s KILL 600 python3 bench.py > gpurun_out/final_default.json 2> gpurun_out/final_default.err; python3 tools/pj.py default < gpurun_out/final_default.json
timeout -s KILL 400 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-config1 --no-serial > gpurun_out/final_dist.json 2> gpurun_out/final_dist.err
grep '^{' gpurun_out/final_dist.json | tail -1 | python3 tools/pj.py dist_default_env
timeout 1200 bash tools/pmc_full.sh r05 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 tables 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 hires 2>&1 | tail -1
timeout 1500 bash tools/profile_round.sh r05 hires 2>&1 | tail -1
timeout -s KILL 600 python3 bench.py --workload hires --no-cpu-baseline > gpurun_out/r5_bench_hires.json 2> gpurun_out/r5_bench_hires.err; python3 tools/pj.py hires < gpurun_out/r5_bench_hires.json
timeout -s KILL 600 python3 bench.py --no-cpu-baseline > gpurun_out/final_default2.json 2> gpurun_out/final_default2.err; python3 tools/pj.py default2 < gpurun_out/final_default2.json
