#!/bin/bash
# k_lat1 built for one wave per SIMD (512 registers: no spills; a workgroup has the compute unit to itself in mode 1)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4w; mkdir -p $O
timeout -s KILL 900 python3 -m pytest tests/test_multiwave.py -m gpu -q -x -s -k "helper" 2>&1 | grep -E "solve of|passed|failed|Error" | tail -5
echo "identical candidates (of 20): $(timeout 300 python3 tools/experiments/r4_helper_dbg.py 2>&1 | grep -c same)"
timeout -s KILL 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-planner --no-serial > $O/b.json 2> $O/b.err; python3 tools/pj.py short < $O/b.json
python3 -c "import json;d=json.load(open('$O/b.json'));print(json.dumps(d['config']['config1_latency'], indent=1))"
