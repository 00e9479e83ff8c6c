#!/bin/bash
# 20-state kernel: the product build against phyml_amd/lib_base (tools/build_variant.sh base on the previous commit), one box:
# parity tests of the product first, then cfg3 and 100 000 patterns, three rounds
export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_fuzz.py tests/test_gpu_replay.py tests/test_gpu_mixture.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for v in lib lib_base; do
  for args in "--workload cfg3_aa_200x10k" "--workload cfg3_aa_200x10k --patterns 100000 --steps 20 --warmup 5"; do
    PHYHIP_LIBDIR=$repo/phyml_amd/$v timeout 200 python bench.py $args --no-cpu-baseline --no-extra --no-call-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['patterns_per_gpu'], 'step_ms', round(d['ms_per_step'],4), 'kernel_us', round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"
  done
done; done
