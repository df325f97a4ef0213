#!/bin/bash
# round 5, first GPU call: the new GPU tests (queued cancel, three-context sharing is CPU, demo exchange) + baseline bench A/B r4 vs tree
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_cancel.py tests/test_cabi.py tests/test_sharding.py tests/test_feasibility.py -m gpu -q -x -s > gpurun_out/r5_first_tests.log 2>&1; grep -E "passed|failed|error|full solve" gpurun_out/r5_first_tests.log | tail -8
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/r5_$tag.json 2> gpurun_out/r5_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r5_$tag.json || tail -3 gpurun_out/r5_$tag.err; }
run first_r4 env TOPAY_LIB=tools/libs/libtopay_r4.so python3 bench.py $A
run first_tree python3 bench.py $A
run first_r4b env TOPAY_LIB=tools/libs/libtopay_r4.so python3 bench.py $A
run first_treeb python3 bench.py $A
