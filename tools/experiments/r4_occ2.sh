#!/bin/bash
# OCC=2 code at one wave per SIMD (grid limited) against the same code at 7 per CU, class N <= 10 alone; then SQ counters
X=$PWD/tools/libs/libtopay_exp.so
echo "== exp over 1.08 (7/CU), N<=10"; TOPAY_LIB=$X timeout 300 python3 tools/gpu_occupancy.py 1024 10
echo "== exp over 0.5 (~4/CU), N<=10"; TOPAY_LIB=$X TOPAY_OVERSUBSCRIBE=0.5 timeout 300 python3 tools/gpu_occupancy.py 1024 10
echo "== exp over 0.75 (~6/CU), N<=10"; TOPAY_LIB=$X TOPAY_OVERSUBSCRIBE=0.75 timeout 300 python3 tools/gpu_occupancy.py 1024 10
echo "== exp over 0.5, 11..15"; TOPAY_LIB=$X TOPAY_OVERSUBSCRIBE=0.5 timeout 300 python3 tools/gpu_occupancy.py 1024 15 11
bash tools/pmc_sq.sh r04a 2>&1 | tail -30
