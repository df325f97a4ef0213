#!/bin/bash
# round 5, final sources (two-loop rewrite, LU loop, store drains, scalar bookkeeping): GPU suite, smoke, default bench lines, then the
# round profile (kernel trace + FETCH / WRITE passes) on these sources
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/final_check.sh 2>&1 | tail -8
timeout 1500 bash tools/profile_round.sh r05 2>&1 | tail -2
bash tools/experiments/r5_stamps.sh 2>&1 | grep -v "^   (" | tail -36
