#!/bin/bash
# 300-step soak of the final build (plain), then the GPU suite once more on the same box
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 1200 python3 bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-config1 --no-planner > gpurun_out/r5_bench_soak300.json 2> gpurun_out/r5_bench_soak300.err; python3 tools/pj.py soak300 < gpurun_out/r5_bench_soak300.json || tail -3 gpurun_out/r5_bench_soak300.err
python3 - <<'PY'
import json, numpy as np
j = json.loads(open("gpurun_out/r5_bench_soak300.json").read().strip().splitlines()[-1])
k = np.array(j["roofline"]["kernel_span_ms_each"])
print("launch spans ms: first 50 steps mean %.0f max %.0f; last 50 mean %.0f max %.0f; overall max %.0f; traffic quoted: %s" % (k[:50].mean(), k[:50].max(), k[-50:].mean(), k[-50:].max(), k.max(), j["roofline"]["traffic"]))
PY
timeout -s KILL 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -2
