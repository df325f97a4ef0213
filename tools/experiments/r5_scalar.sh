#!/bin/bash
# scalar bookkeeping out of the manipulator block: one base address for the parameter block (was: s_getpc + 64-bit add per field
# read, 133 times), 1/K, 10.0 and the seventeen constants of the sine / cosine held in scalar registers (block 9349 -> 8600
# instructions, same arithmetic): hash (must stay 0dbe2e2a1efb9921), evaluation tests, the ESDF-gather kernel alone, and the
# bench, previous build (tools/libs/libtopay_prev11.so) against the tree, interleaved
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5scalar; mkdir -p $O
echo "hash: $(timeout 600 python3 tools/gpu_hashrun.py 2>/dev/null | tail -1)   (before: 0dbe2e2a1efb9921)"
timeout -s KILL 1500 python3 -m pytest tests/test_multiwave.py tests/test_gpu_parity.py tests/test_feasibility.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for v in prev11 tree; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  for k in tables; do echo "K1 $v $k: $(timeout 600 python3 tools/k1_gather.py $k 20 2>/dev/null | tail -1 | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); print(round(j["ms_per_sweep"],4), "ms/sweep", round(j["achieved_GBps"]), "GB/s")')"; done
done
A="--steps 12 --warmup 3 --no-cpu-baseline --no-planner"
for r in 1 2 3; do for v in prev11 tree; do
  if [ $v = tree ]; then unset TOPAY_LIB; else export TOPAY_LIB=$PWD/tools/libs/libtopay_$v.so; fi
  timeout -s KILL 400 python3 bench.py $A > $O/$v$r.json 2> $O/$v$r.err; python3 tools/pj.py $v$r < $O/$v$r.json || tail -3 $O/$v$r.err
  python3 -c "
import json,sys; j=json.load(open('$O/$v$r.json')); c=j['config']['config1_latency']; print('   configs[1] %.1f ms default, %.1f ms helper waves' % (c['solve_ms'], c['solve_ms_latency_mode']))"
done; done
unset TOPAY_LIB
