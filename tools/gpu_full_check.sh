#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/full_gpu_tests.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/full_gpu_tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default_check.json 2> gpurun_out/bench_default_check.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_default_check.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('hbm_frac'), d['extra']['cfg3_aa_200x10k']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
print(json.dumps(d['extra']['call_latency'], indent=1)[:3000])"
