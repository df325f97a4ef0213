#!/bin/bash
# round 5, first profile round on the tree: kernel trace + stats, FETCH / WRITE passes, calibration, default line; SQ counter passes
# of a full-size serial step; the ESDF-gather kernel alone (tables)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 bash tools/profile_round.sh r05 2>&1 | tail -3
timeout 1200 bash tools/pmc_full.sh r05 2>&1 | tail -2
timeout 1500 bash tools/profile_k1.sh r05 tables 2>&1 | tail -5
