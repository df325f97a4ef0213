#!/bin/bash
# cuboids batch (4096 candidates, one map): previous build against the tree, interleaved, product builds
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in 1 2 3; do for v in prev13 tree; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  echo "$v: $(timeout 300 python3 tools/gpu_cuboids_time.py 512 3 2>&1 | tail -1)"
done; done
unset TOPAY_LIB
TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 600 python3 tools/gpu_stamps.py 512 2>&1 | grep -v "^   (" | tail -16
