#!/bin/bash
# Round 3: full GPU suite + default bench (shared maps) + the same with --own-maps + --front-end.
mkdir -p gpurun_out/r3n
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 1500 python -m pytest tests -m gpu -q > gpurun_out/r3n/t.log 2>&1; tail -4 gpurun_out/r3n/t.log
run() { tag=$1; shift; timeout -s KILL 500 "$@" > gpurun_out/r3n/b_$tag.json 2> gpurun_out/r3n/b_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/r3n/b_$tag.json || tail -3 gpurun_out/r3n/b_$tag.err; }
run shared python3 bench.py --no-cpu-baseline
run own python3 bench.py --no-cpu-baseline --own-maps
run frontend python3 bench.py --no-cpu-baseline --front-end
python3 - <<'P'
import json
j=json.loads(open("gpurun_out/r3n/b_frontend.json").read().strip().split("\n")[-1]); print(j["config"]["front_end"], j["config"]["success_fraction"], j["config"]["setup_seconds_untimed"])
j=json.loads(open("gpurun_out/r3n/b_shared.json").read().strip().split("\n")[-1]); print(j["config"]["setup_seconds_untimed"], j["config"]["success_fraction"])
P
