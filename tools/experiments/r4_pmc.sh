#!/bin/bash
bash tools/pmc_full.sh new
TOPAY_LIB=$PWD/tools/libs/libtopay_r3.so bash tools/pmc_full.sh r3
for t in new r3; do echo "=== $t"; python3 tools/pmc_sq_summary.py gpurun_out/pmc_full_$t | grep -A1 "all solve"; done
