#!/bin/bash
# kernel trace of two serial steps: last commit (prev) against the working tree (exp) -- which class launch got longer?
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GPU_MAX_HW_QUEUES=24
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r4v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in prev exp; do
  export TOPAY_LIB=$ROOT/tools/libs/libtopay_$v.so
  rocprofv3 --kernel-trace --stats -d $O/kt_$v -o kt -- python3 $ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-config1 --no-planner --inflight 1 > $O/b_$v.json 2> $O/kt_$v.log
  echo "== $v"; python3 - <<PY
import sqlite3, glob
f = glob.glob("$O/kt_$v/**/*.db", recursive=True)[0]
d = sqlite3.connect(f)
t0 = None
for name, s, e, gx in d.execute("select name, start, end, grid_size_x from kernels where name like '%k_solve%' or name like '%k_lat%' order by start"):
    if t0 is None: t0 = s
    print("  %-14s start %8.1f ms  duration %8.1f ms  grid %d" % (name.split('(')[0], (s - t0) / 1e6, (e - s) / 1e6, gx))
PY
done
