#!/bin/bash
# 20-state kernel against the number of 16-pattern tiles: 256 CUs x 3 workgroup slots = 768 tiles per round.
# cfg3 has 625 tiles: some CUs run 3 workgroups, some 2 -- how much of the distance to the 100 000-pattern figure is that?
export TMPDIR=/tmp
for P in 4096 8192 10000 12288 16384 20480 24576 36864 49152 100000; do
  timeout 200 python bench.py --workload cfg3_aa_200x10k --patterns $P --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; P=$P
print('patterns %6d tiles %5d tiles/768 %.2f  kernel %.1f us  algorithmic frac %.3f  us per tile-round %.1f' % (P, (P+15)//16, ((P+15)//16)/768.0, r['kernel_avg_us'], r['frac'], r['kernel_avg_us']/max(1,-(-((P+15)//16)//768))))"
done
