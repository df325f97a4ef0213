#!/bin/bash
# Round profile on the GPU box (run through gpurun): kernel trace + stats of the bench command, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass), then the counter calibration.  Outputs under gpurun_out/prof_$1.
# usage: tools/profile_round.sh [round tag] [hires]   -- with `hires`: the same three passes on BASELINE configs[4] (4 GB field) only
# the profiler's preloaded library initialises HIP before python starts: the library's own setenv / bench.py's setdefault
# come too late, so the 24 hardware queues of the shipped configuration are asked for here
export GPU_MAX_HW_QUEUES=24
R=${1:-r03}
WL=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R${WL:+_$WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-config1 --inflight 1${WL:+ --workload $WL}"   # strictly serial steps: launches of different steps do not overlap, per-launch figures are well defined
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $ROOT/bench.py $ARGS > $OUT/bench_kt.json 2> $OUT/kt.log
# counter passes serialise the kernels of a step: with the queues shared between the classes the first launch (115
# workgroups) would then solve the whole batch alone, out of L2 -- not the memory behaviour of the real, concurrent run.
# TOPAY_STEAL=0 keeps every class on its own launch for these two passes.
export TOPAY_STEAL=0
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/pmc_write.log
unset TOPAY_STEAL
if [ -n "$WL" ]; then
  python3 $ROOT/bench.py $ARGS > $OUT/bench_plain.json 2>/dev/null
  du -sh $OUT; exit 0
fi
[ -x $ROOT/tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $ROOT/tools/pmc_calib $ROOT/tools/pmc_calib.hip 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $OUT/cal_fetch -o cal -- $ROOT/tools/pmc_calib > $OUT/calib.txt 2> $OUT/cal_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/cal_write -o cal -- $ROOT/tools/pmc_calib >> $OUT/calib.txt 2> $OUT/cal_write.log
# the default (pipelined, three batches in flight) command under the kernel trace, then plain
rocprofv3 --kernel-trace --stats -d $OUT/kt2 -o kt -- python3 $ROOT/bench.py --gpus 1 --steps 6 --warmup 1 --no-cpu-baseline --no-config1 --no-serial > $OUT/bench_kt2.json 2> $OUT/kt2.log
python3 $ROOT/bench.py $ARGS > $OUT/bench_plain.json 2>/dev/null
python3 $ROOT/bench.py > $OUT/bench_default.json 2>/dev/null
find $OUT -name "*.csv" | head -50
# keep the merge-back small: kernel trace CSVs of the bench can be large
find $OUT -name "*.csv" -size +20M -exec gzip {} \;
du -sh $OUT
