#!/bin/bash
# A/B of two builds on one box: TOPAY_LIB=tools/libs/libtopay_base.so against the in-tree library
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/ab_$tag.json || tail -3 gpurun_out/ab_$tag.err; }
A="--steps 16 --warmup 3 --no-cpu-baseline --no-config1"
B=$PWD/tools/libs/${ABLIB:-libtopay_base.so}
echo "== long candidates alone: base"; TOPAY_LIB=$B true
echo "== long candidates alone: new";  true
run base1 env TOPAY_LIB=$B python3 bench.py $A
run new1 python3 bench.py $A
run base2 env TOPAY_LIB=$B python3 bench.py $A
run new2 python3 bench.py $A
for t in base1 new1 base2 new2; do python3 - $t <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1])); c = d["config"]
print(sys.argv[1], "evals/traj %.1f iters %.1f success %.4f serial %s" % (c["mean_evals_per_traj"], c["mean_iters_per_traj"], c["success_fraction"], d["roofline"]["serial_steps"]["ms_per_step"]))
PY
done
