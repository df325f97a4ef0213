#!/bin/bash
run() { tag=$1; shift; timeout -s KILL 300 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 tools/pj.py "$tag" < gpurun_out/ab_$tag.json || tail -3 gpurun_out/ab_$tag.err; }
A="--steps 12 --warmup 3 --no-cpu-baseline --no-config1 --inflight 3"
L=$PWD/tools/libs
for v in "" pf36 pf48; do echo "== alone: ${v:-pf24}"; TOPAY_LIB=${v:+$L/libtopay_$v.so} timeout -s KILL 200 python3 tools/gpu_bigN_time.py 24 40 62; done
run pf24a python3 bench.py $A
run pf36a env TOPAY_LIB=$L/libtopay_pf36.so python3 bench.py $A
run pf48a env TOPAY_LIB=$L/libtopay_pf48.so python3 bench.py $A
run pf24b python3 bench.py $A
run pf36b env TOPAY_LIB=$L/libtopay_pf36.so python3 bench.py $A
run pf48b env TOPAY_LIB=$L/libtopay_pf48.so python3 bench.py $A
