#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench line (+ its detail file)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/full_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/full_gpu_tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_check.json 2> gpurun_out/bench_default_check.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/bench_detail_check.json 2>/dev/null
wc -c gpurun_out/bench_default_check.json; cat gpurun_out/bench_default_check.json
