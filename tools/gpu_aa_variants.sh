#!/bin/bash
# 20-state kernel: builds with compile-time variants (tools/build_variant.sh) against the product, one lease, cfg3 kernel time
export TMPDIR=/tmp
o=gpurun_out/aa_var; mkdir -p $o
for rep in 1 2 3; do
for lib in lib "$@"; do
    PHYHIP_LIBDIR=phyml_amd/$lib timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $o/b.json 2> $o/b.err || tail -3 $o/b.err
    python - "$lib" <<'P'
import json,sys
d=json.load(open('bench_detail.json')); r=d['roofline']
print(sys.argv[1], 'kernel %.1f us step %.1f us'%(r['kernel_avg_us'], d['ms_per_step']*1e3), r['kernel'], 'stored %.1f'%r.get('all_buffers_stored',{}).get('kernel_avg_us',0), 'lnL_err', d.get('lnL_rel_err'))
P
done
done
