#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_fuzz.py tests/test_gpu_replay.py tests/test_gpu_trace.py -x -q -m gpu -k "aa or proteic or 20 or category or fuzz" > gpurun_out/r02c/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r02c/tests.log
for v in skip noskip; do
  if [ $v = noskip ]; then export PHYHIP_AA_NOSKIP=1; else unset PHYHIP_AA_NOSKIP; fi
  timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r02c/bench_cfg3_$v.json 2>gpurun_out/r02c/err_$v.txt
  python -c "import json;d=json.load(open('gpurun_out/r02c/bench_cfg3_$v.json'));print('$v cfg3',d['ms_per_step'],d['roofline']['kernel_avg_us'],d['roofline']['frac'],d['lnL_rel_err'])"
  timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r02c/bench_aa100k_$v.json 2>>gpurun_out/r02c/err_$v.txt
  python -c "import json;d=json.load(open('gpurun_out/r02c/bench_aa100k_$v.json'));print('$v 100k',d['ms_per_step'],d['roofline']['kernel_avg_us'],d['roofline']['frac'])"
done
