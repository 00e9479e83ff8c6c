#!/bin/bash
# first GPU pass of round 2: sharded path tests, full GPU suite, bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
rocm-smi --showid 2>/dev/null | head -5 > gpurun_out/r02a/smi.txt
timeout 1200 python -m pytest tests/test_gpu_shard.py -x -q -m gpu > gpurun_out/r02a/shard_tests.log 2>&1; echo "shard tests rc=$?"
tail -15 gpurun_out/r02a/shard_tests.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_shard.py > gpurun_out/r02a/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/r02a/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_n1.json 2> gpurun_out/r02a/bench_n1.err; echo "bench rc=$?"
cat gpurun_out/r02a/bench_n1.json
PHYHIP_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench_dist1.json 2> gpurun_out/r02a/bench_dist1.err; echo "dist bench rc=$?"
tail -2 gpurun_out/r02a/bench_dist1.json; tail -5 gpurun_out/r02a/bench_dist1.err
