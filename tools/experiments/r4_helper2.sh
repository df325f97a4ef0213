#!/bin/bash
# after the rare-row order fix of the several-waves gradient phase and the helper-wave kernels: hash, tests, default bench (serial
# step of the four-wave classes, configs[1] both ways)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4t; mkdir -p $O
echo "hash product: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
echo "identical candidates (of 20): $(timeout 300 python3 tools/experiments/r4_helper_dbg.py 2>&1 | grep -c same)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py -m gpu -q -x -s 2>&1 | grep -E "solve of|passed|failed|Error" | tail -5
for r in 1 2; do timeout -s KILL 600 python3 bench.py --no-cpu-baseline --no-planner > $O/b$r.json 2> $O/b$r.err; python3 tools/pj.py run$r < $O/b$r.json; done
python3 -c "import json;d=json.load(open('$O/b2.json'));print(json.dumps(d['config']['config1_latency'], indent=1))"
