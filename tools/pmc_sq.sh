#!/bin/bash
# SQ counter passes over one serial bench step (where do the wave cycles of the solve kernels go).  Run through gpurun.
# the profiler's preloaded library initialises HIP before python starts: the library's own setenv / bench.py's setdefault
# come too late, so the 24 hardware queues of the shipped configuration are asked for here
export GPU_MAX_HW_QUEUES=24
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq_${1:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
ARGS="--gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-config1 --inflight 1 --scenarios 256"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/p1 -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/b1.json 2> $OUT/p1.log
timeout -s KILL 300 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM -d $OUT/p2 -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/b2.json 2> $OUT/p2.log
timeout -s KILL 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_IFETCH_LEVEL -d $OUT/p3 -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/b3.json 2> $OUT/p3.log
find $OUT -name "*counter_collection.csv" | while read f; do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")[:28]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "solve" in k or "eval" in k:
        print(k, {a: "%.3g" % b for a, b in sorted(d.items())})
PY
done
grep -i "icache\|IFETCH" $OUT/counters.txt | head -20
tail -3 $OUT/p3.log
