#!/bin/bash
# Experiment: what do the long candidates (N > 32: 72-106 KB of LDS per workgroup, the longest solves) cost the rest of
# the batch?  Same batch with them left out; compare the work rate (roofline.achieved = algorithmic bytes of the solved
# candidates / time), not trajectories/s.
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 - $tag <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1])); r = d["roofline"]
    print(sys.argv[1], "ms/step %.1f  achieved %.0f GB/s  algorithmic TB %.3f  serial ms %s" % (d["ms_per_step"], r["achieved"], r["algorithmic_bytes_per_step"] / 1e12, (r.get("serial_steps") or {}).get("ms_per_step")))
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
}
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1"
L=$PWD/tools/libs/libtopay_skip.so
run all env TOPAY_LIB=$L python3 bench.py $A
run le42 env TOPAY_LIB=$L TOPAY_EXPERIMENT_SKIP_ABOVE=42 python3 bench.py $A
run le32 env TOPAY_LIB=$L TOPAY_EXPERIMENT_SKIP_ABOVE=32 python3 bench.py $A
run le21 env TOPAY_LIB=$L TOPAY_EXPERIMENT_SKIP_ABOVE=21 python3 bench.py $A
run all2 env TOPAY_LIB=$L python3 bench.py $A
