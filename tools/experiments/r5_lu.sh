#!/bin/bash
# LU pivot loop with all four LDS reads ahead of the division (one round trip per pivot): hash (must stay 0dbe2e2a1efb9921),
# planning call by phase, configs[1] latency
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (before: 0dbe2e2a1efb9921)"
TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 600 python3 tools/gpu_stamps_cfg1.py 2>&1 | grep -v "^     (" | tail -34
