#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_fuzz.py tests/test_gpu_replay.py tests/test_gpu_trace.py -x -q -m gpu > gpurun_out/r02b/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02b/tests.log
for gen in 2 1; do
  PHYHIP_AA_GEN=$gen timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r02b/bench_cfg3_gen$gen.json 2>gpurun_out/r02b/err_$gen.txt
  python -c "import json;d=json.load(open('gpurun_out/r02b/bench_cfg3_gen$gen.json'));print('gen$gen cfg3',d['ms_per_step'],d['roofline']['kernel_avg_us'],d['roofline']['frac'],d['lnL_rel_err'])"
  PHYHIP_AA_GEN=$gen timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r02b/bench_aa100k_gen$gen.json 2>>gpurun_out/r02b/err_$gen.txt
  python -c "import json;d=json.load(open('gpurun_out/r02b/bench_aa100k_gen$gen.json'));print('gen$gen 100k',d['ms_per_step'],d['roofline']['kernel_avg_us'],d['roofline']['frac'])"
done
timeout 1200 python -m pytest tests/test_gpu_shard.py tests/test_gpu_cfg5.py -x -q -m gpu > gpurun_out/r02b/tests2.log 2>&1; echo "tests2 rc=$?"
tail -5 gpurun_out/r02b/tests2.log
