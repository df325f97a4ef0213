#!/bin/bash
# where the solver's time between two evaluations goes (diagnostics build with the LSTAMP sub-stamps), one and two waves per SIMD
export HSA_ENABLE_IPC_MODE_LEGACY=0
for S in 128 512; do echo "== S=$S"; TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 900 python3 tools/gpu_stamps.py $S 2>&1 | tail -24 | head -20; done
