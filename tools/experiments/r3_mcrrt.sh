#!/bin/bash
# Round 3: the device MCRRT -- GPU parity tests, then the time of a benchmark-sized batch (hard kill time-outs).
mkdir -p gpurun_out/r3m
timeout -s KILL 900 python -m pytest tests/test_mcrrt.py -m gpu -x -q -s > gpurun_out/r3m/t.log 2>&1; tail -8 gpurun_out/r3m/t.log
timeout -s KILL 600 python3 tools/gpu_mcrrt_time.py 2>&1 | tail -8
