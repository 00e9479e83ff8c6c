#!/bin/bash
# round 5: the whole GPU suite with virtual buffers on by default, then lane groups at cfg2 and the 1 M-pattern line
export TMPDIR=/tmp
o=gpurun_out/r5b; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu -x > $o/full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -20; tail -30 $o/full.log | head -60
for g in 1 2; do
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_NT_GROUPS=$g timeout 300 python bench.py --workload cfg2_nt_100x50k --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $o/bench_cfg2_g$g.json 2> $o/bench_cfg2_g$g.err
  python -c "
import json; d=json.load(open('$o/bench_cfg2_g$g.json')); r=d['roofline']; print('cfg2 groups $g ms/step %.4f kernel %.1f stored %s' % (d['ms_per_step'], r['kernel_avg_us'], r.get('all_buffers_stored',{}).get('kernel_avg_us')))"
done
timeout 300 python bench.py --workload cfg4_nt_100x1M --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $o/bench_1M.json 2> $o/bench_1M.err
python -c "
import json; d=json.load(open('$o/bench_1M.json')); r=d['roofline']; print('1M ms/step', d['ms_per_step'], 'kernel us', r['kernel_avg_us'], 'value', d['value'], r.get('all_buffers_stored'))"
