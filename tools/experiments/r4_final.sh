#!/bin/bash
# Final measurements of round 4 in one gpurun call: new GPU test, round profile (kernel trace, FETCH / WRITE passes, default
# bench), SQ counters, the ESDF-gather kernel alone, bench variants (hires, front end, RCCL path), code-object report inputs.
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4final; mkdir -p $O
echo "== wall-clock budget test"; timeout 600 python3 -m pytest tests/test_cancel.py -x -q -m gpu -s 2>&1 | grep -v "^\[" | tail -4
echo "== profile round"; timeout 1800 bash tools/profile_round.sh r04 2>&1 | tail -2
echo "== pmc sq (small)"; timeout 900 bash tools/pmc_sq.sh r04 > $O/pmc_sq.log 2>&1; tail -1 $O/pmc_sq.log
echo "== pmc full"; timeout 1200 bash tools/pmc_full.sh r04 2>&1 | tail -1
echo "== k1 tables"; timeout 900 bash tools/profile_k1.sh r04 tables > $O/k1_tables.log 2>&1; tail -3 $O/k1_tables.log
run() { tag=$1; shift; timeout -s KILL 600 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
run hires python3 bench.py --workload hires --steps 12 --warmup 3 --no-cpu-baseline
run front_end python3 bench.py --front-end --steps 12 --warmup 3 --no-cpu-baseline --no-config1
timeout -s KILL 500 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-config1 --no-serial > $O/b_rccl.json 2> $O/b_rccl.err
grep '^{' $O/b_rccl.json | tail -1 > $O/b_rccl_line.json; python3 tools/pj.py rccl_path < $O/b_rccl_line.json
run serial python3 bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-config1 --no-planner
run default python3 bench.py
