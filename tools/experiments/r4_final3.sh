#!/bin/bash
# profile round on the final sources + the ESDF-gather kernel on the cached maps and on the 4 GB field
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4final3; mkdir -p $O
echo "== profile round"; timeout 1800 bash tools/profile_round.sh r04 2>&1 | tail -1
echo "== k1 tables"; timeout 900 bash tools/profile_k1.sh r04 tables > $O/k1_tables.log 2>&1; tail -1 $O/k1_tables.log
echo "== k1 hires"; timeout 1500 bash tools/profile_k1.sh r04 hires > $O/k1_hires.log 2>&1; tail -1 $O/k1_hires.log
timeout -s KILL 600 python3 bench.py > $O/default.json 2> $O/default.err; python3 tools/pj.py default < $O/default.json
