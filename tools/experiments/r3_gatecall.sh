#!/bin/bash
# Round 3: in-solve gate as a call (k_solve1 39 instead of 161 spilled VGPRs) -- parity subset + default bench, twice.
mkdir -p gpurun_out/r3k
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_cancel.py tests/test_feasibility.py -m gpu -x -q > gpurun_out/r3k/t.log 2>&1; tail -3 gpurun_out/r3k/t.log
timeout -s KILL 120 python3 tools/gpu_hashrun.py 2>&1 | tail -2
for i in 1 2; do timeout -s KILL 400 python3 bench.py --no-cpu-baseline > gpurun_out/r3k/b$i.json 2> gpurun_out/r3k/b$i.err; python3 tools/pj.py default$i < gpurun_out/r3k/b$i.json; done
python3 - <<'P'
import json
j=json.loads(open("gpurun_out/r3k/b2.json").read().strip().split("\n")[-1]); r=j["roofline"]
print({k: r[k] for k in ("slot_seconds_per_step","work_ms_per_slot","slot_utilisation","frac")}, j["config"]["planner_semantics"])
P
