#!/bin/bash
# last call of round 4: GPU suite + smoke, round profile on the final sources (traffic figure with their source_sha), default bench
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4final5; mkdir -p $O
timeout -s KILL 1800 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
timeout -s KILL 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "== profile round"; timeout 1800 bash tools/profile_round.sh r04 2>&1 | tail -1
timeout -s KILL 600 python3 bench.py > $O/default.json 2> $O/default.err; python3 tools/pj.py default < $O/default.json
python3 -c "import json;d=json.load(open('$O/default.json'));print('config1', d['config'].get('config1_latency')); print('traffic', d['roofline'].get('traffic'))"
