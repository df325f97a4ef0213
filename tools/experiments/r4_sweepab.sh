#!/bin/bash
# band_sweep with block-ahead loads against the committed build (hash, interleaved bench, config1 latency) + stamps with
# the exposed wait of the ESDF gathers
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4m; mkdir -p $O
echo "hash product: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
echo "hash prev:    $(TOPAY_LIB=tools/libs/libtopay_prev.so timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
for r in 1 2; do for v in prev exp; do
  TOPAY_LIB=tools/libs/libtopay_$v.so timeout -s KILL 600 python3 bench.py --no-cpu-baseline --no-planner > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json
  python3 -c "import json;d=json.load(open('$O/$v$r.json'));print('   config1', d.get('config1_latency'))"
done; done
echo "== stamps"; TOPAY_LIB=tools/libs/libtopay_stamps.so timeout 900 python3 tools/gpu_stamps.py 512 2>&1 | tail -22
