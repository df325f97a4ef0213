#!/bin/bash
# round 5, final sources (host cancel flag polled at every eighth evaluation): GPU suite, smoke, default bench lines, the round profile
# and the remaining profile pieces on these sources
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/final_check.sh 2>&1 | tail -8
timeout 1500 bash tools/profile_round.sh r05 2>&1 | tail -1
timeout 1200 bash tools/pmc_full.sh r05 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 tables 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 hires 2>&1 | tail -1
timeout 1500 bash tools/profile_round.sh r05 hires 2>&1 | tail -1
timeout -s KILL 600 python3 bench.py --workload hires --no-cpu-baseline > gpurun_out/r5_bench_hires.json 2> gpurun_out/r5_bench_hires.err; python3 tools/pj.py hires < gpurun_out/r5_bench_hires.json
timeout -s KILL 600 python3 bench.py --front-end --no-cpu-baseline > gpurun_out/r5_bench_front_end.json 2> gpurun_out/r5_bench_front_end.err; python3 tools/pj.py front_end < gpurun_out/r5_bench_front_end.json
timeout -s KILL 600 python3 bench.py --no-cpu-baseline > gpurun_out/final_default2.json 2> gpurun_out/final_default2.err; python3 tools/pj.py default2 < gpurun_out/final_default2.json
