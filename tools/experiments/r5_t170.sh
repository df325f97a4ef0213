#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 1200 python3 -m pytest tests/test_multiwave.py -m gpu -q -x -s 2>&1 | grep "capped solve\|passed\|failed"
