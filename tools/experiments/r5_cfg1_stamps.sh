#!/bin/bash
# planning call (configs[1]) by phase, default kernels and helper waves (diagnostics build)
export HSA_ENABLE_IPC_MODE_LEGACY=0
TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 600 python3 tools/gpu_stamps_cfg1.py 2>&1 | tail -60
