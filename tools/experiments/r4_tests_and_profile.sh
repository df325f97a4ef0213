#!/bin/bash
echo "== new gpu tests"; timeout 1500 python3 -m pytest tests/test_sharding.py tests/test_outputs_gpu.py tests/test_feasibility.py tests/test_cabi.py -x -q -m gpu -s 2>&1 | grep -v "^\[" | tail -25
echo "== profile round"; timeout 2400 bash tools/profile_round.sh r04 2>&1 | tail -5
