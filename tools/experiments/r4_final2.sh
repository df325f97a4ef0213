#!/bin/bash
# Last gpurun call of round 4: full GPU suite + smoke, the round profile on the final sources (its traffic figure carries their
# source_sha), an 80-step soak, the default bench line.
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4final2; mkdir -p $O
timeout -s KILL 1800 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
timeout -s KILL 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "== profile round"; timeout 1800 bash tools/profile_round.sh r04 2>&1 | tail -1
echo "== soak"; timeout -s KILL 600 python3 bench.py --steps 80 --warmup 3 --no-cpu-baseline --no-config1 --no-planner > $O/soak.json 2> $O/soak.err; python3 tools/pj.py soak80 < $O/soak.json
timeout -s KILL 600 python3 bench.py > $O/default.json 2> $O/default.err; python3 tools/pj.py default < $O/default.json
