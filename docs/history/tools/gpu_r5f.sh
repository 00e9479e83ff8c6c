#!/bin/bash
# round 5: whole suite after (materialise-all at the first short read, grouped class-axis mixtures, 20-state in-step children); mixture groups bench
export TMPDIR=/tmp
o=gpurun_out/r5f; mkdir -p $o
timeout 1800 python -m pytest tests -q -m gpu > $o/full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -20
for k in 10 60; do timeout 300 python tools/bench_mixture_groups.py $k 2000 100 2>&1 | tail -1 | tee $o/mixture_groups_$k.json; done
timeout 300 python tools/bench_mixture_groups.py 10 20000 100 2>&1 | tail -1 | tee $o/mixture_groups_10_20000.json
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$o/bench_default.json'))
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], r['kernel_avg_us'], r['frac'], r.get('all_buffers_stored'))
e=d['extra']; c3=e['cfg3_aa_200x10k']; print('cfg3', c3['value'], c3['ms_per_step'], c3['roofline']['kernel_avg_us'], c3['roofline']['frac'], c3.get('mfma'))
print('cfg4', e['cfg4_nt_100x1M_one_gpu'])
print(json.dumps(e['call_latency'], indent=1)[:2500]); print(d['cpu_baseline'])"
