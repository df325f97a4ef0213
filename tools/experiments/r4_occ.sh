#!/bin/bash
R3=$PWD/tools/libs/libtopay_r3.so
echo "== new, N<=10 only"; timeout 300 python3 tools/gpu_occupancy.py 1024 10
echo "== r3, N<=10 only"; TOPAY_LIB=$R3 timeout 300 python3 tools/gpu_occupancy.py 1024 10
echo "== new, 11..15"; timeout 300 python3 tools/gpu_occupancy.py 1024 15 11
echo "== r3, 11..15"; TOPAY_LIB=$R3 timeout 300 python3 tools/gpu_occupancy.py 1024 15 11
echo "== new, all"; timeout 300 python3 tools/gpu_occupancy.py 1024
echo "== r3, all"; TOPAY_LIB=$R3 timeout 300 python3 tools/gpu_occupancy.py 1024
