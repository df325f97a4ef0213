mkdir -p gpurun_out/r3a
./tools/wavesum_probe > gpurun_out/r3a/wavesum.txt 2>&1
for r in "4 10" "11 15" "16 21" "22 32" "33 42" "43 64"; do
  echo "== N in $r" >> gpurun_out/r3a/stamps.txt
  timeout -s KILL 300 python3 tools/gpu_stamps_bigN.py $r >> gpurun_out/r3a/stamps.txt 2>&1
done
timeout -s KILL 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench_base.json 2> gpurun_out/r3a/bench_base.err
cat gpurun_out/r3a/wavesum.txt; tail -5 gpurun_out/r3a/stamps.txt; python3 tools/pj.py base < gpurun_out/r3a/bench_base.json
