export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 1500 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/final_tests.log | tail -3
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -s KILL 400 env TOPAY_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-config1 --no-serial > gpurun_out/final_dist.json 2> gpurun_out/final_dist.err
grep '^{' gpurun_out/final_dist.json | tail -1 | python3 tools/pj.py dist_default_env
timeout -s KILL 600 python3 bench.py > gpurun_out/final_default.json 2> gpurun_out/final_default.err
python3 tools/pj.py default < gpurun_out/final_default.json
