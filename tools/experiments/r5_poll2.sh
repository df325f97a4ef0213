#!/bin/bash
# after the poll change: do the slow phases still appear?  150-step soaks (plain and through the RCCL code path) and the 4096-candidate
# cuboids launch six times, then the same launch six times with the previous build
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/soak.sh 150 2>&1 | tail -4
echo "tree:   $(timeout 300 python3 tools/gpu_cuboids_time.py 512 6 2>&1 | tail -1)"
echo "prev14: $(TOPAY_LIB=$PWD/tools/libs/libtopay_prev14.so timeout 300 python3 tools/gpu_cuboids_time.py 512 6 2>&1 | tail -1)"
echo "tree:   $(timeout 300 python3 tools/gpu_cuboids_time.py 512 6 2>&1 | tail -1)"
