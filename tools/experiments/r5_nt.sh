#!/bin/bash
# non-temporal hint on the history loads of the two-loop recursion: interleaved bench against the same sources without it
# (experiments build, -DTOPAY_HIST_AUX=0), then FETCH_SIZE / WRITE_SIZE of a serial step both ways
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GPU_MAX_HW_QUEUES=24
O=$PWD/gpurun_out/r5nt; mkdir -p $O
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)"
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --no-planner"
for r in 1 2 3; do for v in nont nt; do
  if [ $v = nt ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
done; done
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-config1 --no-planner --inflight 1"
export TOPAY_STEAL=0
for v in nont nt; do
  if [ $v = nt ]; then unset TOPAY_LIB; else export TOPAY_LIB=$ROOT/tools/libs/libtopay_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 400 rocprofv3 --pmc $c -d $O/pmc_${v}_$c -o pmc -- python3 $ROOT/bench.py $ARGS > $O/pmc_${v}_$c.json 2> $O/pmc_${v}_$c.log
    python3 - $O/pmc_${v}_$c/pmc_results.db $v $c <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tot = 0.0
for name, v in c.execute("select kernel_name, sum(value) from counters_collection group by kernel_name"):
    if name.startswith(("k_solve", "k_long")): tot += v
print(sys.argv[2], sys.argv[3], "sum over the solve kernels of 3 steps: %.3e (raw counter units)" % tot)
PY
  done
done
