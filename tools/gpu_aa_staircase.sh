#!/bin/bash
# 20-state kernel: kernel time against wave-tiles per CU (patterns = 1024 k -> 256 k tiles -> k consumer waves on each of 256 CUs):
# a staircase with steps at k = 5, 9, 13 says "serial per SIMD" (the busiest SIMD holds ceil(k / 4) consumers), a line says "a per-CU resource"
export TMPDIR=/tmp
o=gpurun_out/aa_stair; mkdir -p $o
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
    timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns $((1024*k)) --steps 15 --warmup 4 --no-cpu-baseline --no-extra --no-companion > $o/b.json 2> $o/b.err || tail -3 $o/b.err
    python - "$k" <<'P'
import json,sys
d=json.load(open('bench_detail.json')); r=d['roofline']
print('tiles/CU',sys.argv[1], 'kernel %.1f us'%r['kernel_avg_us'], r['kernel'])
P
done
