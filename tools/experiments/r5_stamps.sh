#!/bin/bash
# per-phase shader-clock stamps of the round-5 code at one and at two resident waves per SIMD (diagnostics build)
export HSA_ENABLE_IPC_MODE_LEGACY=0
for S in 128 512; do echo "== S=$S"; TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 900 python3 tools/gpu_stamps.py $S 2>&1 | tail -24; done
