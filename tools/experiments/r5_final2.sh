#!/bin/bash
# round 5, final sources (second time, after the pair gathers): GPU suite + smoke + RCCL path + default line (tools/final_check.sh), then the profile round, the SQ counter
# passes, the ESDF-gather kernel alone on both workloads, configs[4] bench and its traffic passes
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/final_check.sh 2>&1 | tail -8
timeout 2400 bash tools/profile_round.sh r05 2>&1 | tail -2
timeout 1200 bash tools/pmc_full.sh r05 2>&1 | tail -1
timeout 1500 bash tools/profile_k1.sh r05 tables 2>&1 | tail -2
timeout 1500 bash tools/profile_k1.sh r05 hires 2>&1 | tail -2
timeout 1500 bash tools/profile_round.sh r05 hires 2>&1 | tail -2
timeout -s KILL 600 python3 bench.py --workload hires --no-cpu-baseline > gpurun_out/r5_bench_hires.json 2> gpurun_out/r5_bench_hires.err; python3 tools/pj.py hires < gpurun_out/r5_bench_hires.json
timeout -s KILL 900 python3 bench.py --steps 80 --warmup 3 --no-cpu-baseline --no-config1 --no-planner > gpurun_out/r5_bench_soak80.json 2> gpurun_out/r5_bench_soak80.err; python3 tools/pj.py soak80 < gpurun_out/r5_bench_soak80.json
timeout -s KILL 600 python3 bench.py --front-end --no-cpu-baseline --no-config1 > gpurun_out/r5_bench_front_end.json 2> gpurun_out/r5_bench_front_end.err; python3 tools/pj.py front_end < gpurun_out/r5_bench_front_end.json
