#!/bin/bash
# clocks and power under the solve: r3 library (one wave per SIMD) against the two-waves build
O=gpurun_out/r4c; mkdir -p $O
R3=$PWD/tools/libs/libtopay_r3.so
poll() { while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temp" | tr '\n' ' '; echo; sleep 0.5; done; }
A="--steps 10 --warmup 3 --no-cpu-baseline --no-planner --no-config1 --no-serial"
rocm-smi --showclocks --showpower 2>&1 | head -30
poll > $O/smi_r3.txt & P=$!
TOPAY_LIB=$R3 timeout 300 python3 bench.py $A > $O/b_r3.json 2> $O/b_r3.err; kill $P
sleep 2
poll > $O/smi_new.txt & P=$!
timeout 300 python3 bench.py $A > $O/b_new.json 2> $O/b_new.err; kill $P
python3 tools/pj.py r3 < $O/b_r3.json; python3 tools/pj.py new < $O/b_new.json
echo "== r3 smi (every 4th sample)"; awk 'NR%4==0' $O/smi_r3.txt | tail -12
echo "== new smi"; awk 'NR%4==0' $O/smi_new.txt | tail -12
