#!/bin/bash
# round 5, final sources (170-piece class): the whole GPU suite once more, then the remaining profile pieces on these sources
# (SQ counters, the ESDF-gather kernel alone, configs[4] traffic and bench line)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 1800 python3 -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; grep -E "passed|failed" gpurun_out/final_tests.log | tail -2
timeout 1200 bash tools/pmc_full.sh r05 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 tables 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 hires 2>&1 | tail -1
timeout 1500 bash tools/profile_round.sh r05 hires 2>&1 | tail -1
timeout -s KILL 600 python3 bench.py --workload hires --no-cpu-baseline > gpurun_out/r5_bench_hires.json 2> gpurun_out/r5_bench_hires.err; python3 tools/pj.py hires < gpurun_out/r5_bench_hires.json
