#!/bin/bash
# round 5, final sources (third time: TOPAY_MAX_N 170, k_long14 / k_eval4w4): hash, GPU suite + smoke + RCCL path + default line, profile round
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (before: 0dbe2e2a1efb9921)"
bash tools/final_check.sh 2>&1 | tail -8
grep -E "passed|failed" gpurun_out/final_tests.log | tail -2
timeout 2400 bash tools/profile_round.sh r05 2>&1 | tail -2
timeout 300 python3 tools/gpu_bigN_time.py 100.0 133.0 177.0 2>&1 | tail -4
