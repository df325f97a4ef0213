#!/bin/bash
# rocprofv3 passes over the ESDF-gather kernel alone (tools/k1_gather.py): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in separate passes.  Run through gpurun; outputs under gpurun_out/k1_$1_{tables,hires}.
# the profiler's preloaded library initialises HIP before python starts: the library's own setenv / bench.py's setdefault
# come too late, so the 24 hardware queues of the shipped configuration are asked for here
export GPU_MAX_HW_QUEUES=24
R=${1:-r02}; KIND=${2:-tables}; REPS=${3:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/k1_${R}_$KIND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 900 python3 $ROOT/tools/k1_gather.py $KIND $REPS > $OUT/plain.json 2> $OUT/plain.err
timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $ROOT/tools/k1_gather.py $KIND $REPS > $OUT/kt.json 2> $OUT/kt.log
timeout -s KILL 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python3 $ROOT/tools/k1_gather.py $KIND $REPS > $OUT/fetch.json 2> $OUT/fetch.log
timeout -s KILL 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- python3 $ROOT/tools/k1_gather.py $KIND $REPS > $OUT/write.json 2> $OUT/write.log
cd $ROOT
python3 tools/summarize_k1.py $OUT
