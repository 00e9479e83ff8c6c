#!/bin/bash
# round 5: lanes per pattern at the shard sizes of the scaling run (1 M patterns over 8 / 4 / 2 GPUs) with virtual buffers and two wave shapes
export TMPDIR=/tmp
o=gpurun_out/r5m; mkdir -p $o
for P in 125000 250000 500000; do
 for g in 1 2; do
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_NT_GROUPS=$g timeout 200 python bench.py --workload cfg4_nt_100x125k --patterns $P --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-companion > $o/b_${P}_g$g.json 2> $o/b_${P}_g$g.err
  python -c "
import json; d=json.load(open('$o/b_${P}_g$g.json')); r=d['roofline']; print('patterns $P groups $g: kernel %.1f us, step %.1f us' % (r['kernel_avg_us'], d['ms_per_step']*1e3))"
 done
done
