#!/bin/bash
# helper-wave kernels (topay_set_latency_mode): GPU tests of the several-waves evaluation and of the helper kernels, the result hash
# of the default kernels, BASELINE configs[1] both ways, which candidates differ (none)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4r; mkdir -p $O
echo "hash product: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
timeout 300 python3 tools/experiments/r4_helper_dbg.py 2>&1 | grep -c same
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py -m gpu -q -x -s 2>&1 | grep -E "solve of|passed|failed|Error" | tail -5
timeout -s KILL 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-planner --no-serial > $O/b.json 2> $O/b.err; python3 tools/pj.py short < $O/b.json
python3 -c "import json;d=json.load(open('$O/b.json'));print(json.dumps(d['config']['config1_latency'], indent=1))"
