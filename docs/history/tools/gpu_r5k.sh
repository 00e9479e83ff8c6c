#!/bin/bash
# round 5: two wave shapes in one traversal launch -- the whole suite, then cfg2 / cfg5-size with and without
export TMPDIR=/tmp
o=gpurun_out/r5k; mkdir -p $o
timeout 1800 python -m pytest tests -q -m gpu > $o/full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -20
for mx in 1 0; do
  for P in 50000 40000 20000 100000; do
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_NT_MIXED=$mx timeout 300 python bench.py --workload cfg2_nt_100x50k --patterns $P --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-companion > $o/b_$mx_$P.json 2> $o/b_$mx_$P.err
  python -c "
import json; d=json.load(open('$o/b_$mx_$P.json')); r=d['roofline']; print('mixed $mx patterns $P: kernel %.1f us, step %.1f us, lnL %.10f' % (r['kernel_avg_us'], d['ms_per_step']*1e3, d['lnL']))"
  done
done
