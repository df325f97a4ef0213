#!/bin/bash
# Final measurements of round 4 on the final sources (LU factors streamed, one LDS layout): full GPU suite + smoke, round profile
# (kernel trace, FETCH / WRITE passes, default bench), SQ counters, the ESDF-gather kernel alone on both fields, bench variants.
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4final4; mkdir -p $O
timeout -s KILL 1800 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
timeout -s KILL 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "== profile round"; timeout 1800 bash tools/profile_round.sh r04 2>&1 | tail -1
echo "== pmc full"; timeout 1200 bash tools/pmc_full.sh r04 2>&1 | tail -1
echo "== k1 tables"; timeout 900 bash tools/profile_k1.sh r04 tables > $O/k1_tables.log 2>&1; tail -1 $O/k1_tables.log
echo "== k1 hires"; timeout 1500 bash tools/profile_k1.sh r04 hires > $O/k1_hires.log 2>&1; tail -1 $O/k1_hires.log
run() { tag=$1; shift; timeout -s KILL 600 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 tools/pj.py "$tag" < $O/b_$tag.json || tail -3 $O/b_$tag.err; }
run hires python3 bench.py --workload hires --steps 12 --warmup 3 --no-cpu-baseline
run front_end python3 bench.py --front-end --steps 12 --warmup 3 --no-cpu-baseline --no-config1
timeout -s KILL 500 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-config1 --no-serial > $O/b_rccl.json 2> $O/b_rccl.err
grep '^{' $O/b_rccl.json | tail -1 > $O/b_rccl_line.json; python3 tools/pj.py rccl_path < $O/b_rccl_line.json
run serial python3 bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-config1 --no-planner
run soak80 python3 bench.py --steps 80 --warmup 3 --no-cpu-baseline --no-config1 --no-planner
run default python3 bench.py
python3 -c "import json;d=json.load(open('$O/b_default.json'));print('config1', d['config'].get('config1_latency')); print('traffic', d['roofline'].get('traffic'))"
