#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_cabi.py tests/test_sharding.py -m gpu -q -x > gpurun_out/r5_t2_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5_t2_tests.log
