#!/bin/bash
# 300 pipelined steps of the default command on the final build (2.46 million trajectories)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 900 python3 bench.py --gpus 1 --steps 300 --warmup 2 --no-cpu-baseline --no-config1 --no-serial --no-planner > gpurun_out/soak300.json 2> gpurun_out/soak300.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/soak300.json") if l.startswith("{")][-1]); k = d["roofline"]["kernel_span_ms_each"]
q = lambda a: (min(a), sorted(a)[len(a)//2], max(a))
print("steps", d["steps"], "value %.0f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "spans min/median/max first 50: %.0f %.0f %.0f  last 50: %.0f %.0f %.0f" % (q(k[:50]) + q(k[-50:])))
PY
