#!/bin/bash
# Soak: many pipelined steps on the default settings, plain and through the RCCL code path (one GPU)
export HSA_ENABLE_IPC_MODE_LEGACY=0
A="--gpus 1 --steps ${1:-80} --warmup 2 --no-cpu-baseline --no-config1 --no-serial"
timeout -s KILL 600 python3 bench.py $A > gpurun_out/soak_plain.json 2> gpurun_out/soak_plain.err; python3 tools/pj.py soak_plain < gpurun_out/soak_plain.json || tail -3 gpurun_out/soak_plain.err
timeout -s KILL 600 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29523 bench.py $A > gpurun_out/soak_dist.json 2> gpurun_out/soak_dist.err; grep '^{' gpurun_out/soak_dist.json | tail -1 | python3 tools/pj.py soak_dist || tail -3 gpurun_out/soak_dist.err
python3 - <<'PY'
import json
for t in ("plain", "dist"):
    try:
        d = json.loads([l for l in open("gpurun_out/soak_%s.json" % t) if l.startswith("{")][-1]); k = d["roofline"]["kernel_span_ms_each"]
        print(t, "steps", d["steps"], "value %.0f" % d["value"], "spans min/median/max %.0f %.0f %.0f" % (min(k), sorted(k)[len(k)//2], max(k)))
    except Exception as ex:
        print(t, "no line:", ex)
PY
