#!/bin/bash
# Round 3: whole GPU suite + smoke + default bench line (one gpurun call).
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3e
timeout -s KILL 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r3e/gpu_suite.log 2>&1; tail -4 gpurun_out/r3e/gpu_suite.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -s KILL 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3e/bench_default.json 2> gpurun_out/r3e/bench_default.err
python3 tools/pj.py default < gpurun_out/r3e/bench_default.json
