#!/bin/bash
# usage: tools/bench_sweep.sh "<tag>" "<bench args>" ...   (pairs)  -- runs bench.py with a hard kill timeout and prints value / ms per step
while [ $# -ge 2 ]; do
  tag=$1; args=$2; shift 2
  s=$(date +%s)
  timeout -s KILL 400 python bench.py $args > gpurun_out/r2_bench_$tag.json 2> gpurun_out/r2_bench_$tag.err
  rc=$?
  e=$(date +%s)
  python - "$tag" $rc $((e-s)) <<'PY'
import json, sys
tag, rc, wall = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.load(open("gpurun_out/r2_bench_%s.json" % tag))
    r = d["roofline"]
    print(tag, "rc", rc, "wall", wall, "s value %.0f ms/step %.1f spans %s serial %s" % (d["value"], d["ms_per_step"], [int(x) for x in r["kernel_span_ms_each"]],
          (r.get("serial_steps") or {}).get("ms_per_step")))
except Exception as ex:
    print(tag, "rc", rc, "wall", wall, "no json:", ex)
PY
done
