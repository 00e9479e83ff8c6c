#!/bin/bash
# A/B of the step-level switches (valid results): final sum fused into the traversal vs separate launch
export TMPDIR=/tmp
run() { PHYHIP_SPLIT_REDUCE=$1 timeout 200 python bench.py --workload $2 $3 --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"; }
for sr in 0 1; do
  echo "cfg2 split_reduce=$sr step_us kernel_us = $(run $sr cfg2_nt_100x50k)"
  echo "nt 125k split_reduce=$sr = $(run $sr cfg2_nt_100x50k '--patterns 125000')"
  echo "nt 1M split_reduce=$sr = $(run $sr cfg2_nt_100x50k '--patterns 1000000')"
  echo "cfg3 split_reduce=$sr = $(run $sr cfg3_aa_200x10k)"
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_mixture.py -x -q -m gpu 2>&1 | tail -3
