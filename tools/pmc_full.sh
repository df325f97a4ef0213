#!/bin/bash
# Counter passes over one strictly serial bench step at full size (SQ issue / wait, instruction cache, instruction mix).  The
# TCP / TCC counters are left out: a pass with them takes more than five minutes on this workload.
# usage: tools/pmc_full.sh <tag>   (TOPAY_LIB selects the build).  Run through gpurun; summary: tools/pmc_sq_summary.py
export GPU_MAX_HW_QUEUES=24
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_full_${1:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-config1 --no-planner --inflight 1"
pass() { n=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" -d $OUT/p$n -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/b$n.json 2> $OUT/p$n.log; }
pass 1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
pass 2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQC_DCACHE_REQ SQC_DCACHE_MISSES
pass 3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
tail -2 $OUT/p3.log
