#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_full.log 2>&1; grep -E "passed|failed" gpurun_out/tests_full.log | tail -2
for hs in 1 0; do
  echo "host_sum=$hs"
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_spr.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg5 us/candidate', round(d['us_per_candidate'],2), 'full ms', round(d['full_both_sides_Lk_ms'],2))"
  PHYHIP_HOST_SUM=$hs timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg2 step_us', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"
  PHYHIP_HOST_SUM=$hs timeout 200 python bench.py --workload cfg3_aa_200x10k --steps 100 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg3 step_us', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"
  PHYHIP_HOST_SUM=$hs timeout 200 python bench.py --patterns 1000000 --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  1M step_us', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_avg_us'],1))"
done
