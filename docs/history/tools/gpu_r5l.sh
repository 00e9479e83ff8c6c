#!/bin/bash
# round 5: consumer waves per workgroup of the 20-state kernel at cfg3 on the final kernel (in-step children, virtual buffers); the default bench line
export TMPDIR=/tmp
o=gpurun_out/r5l; mkdir -p $o
for nw in 10 8 9 11 12 13 15; do
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_AA_NW=$nw timeout 200 python bench.py --workload cfg3_aa_200x10k --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-companion > $o/aa_nw$nw.json 2> $o/aa_nw$nw.err
  python -c "
import json; d=json.load(open('$o/aa_nw$nw.json')); r=d['roofline']; print('aa_nw $nw: kernel %.1f us, step %.1f us' % (r['kernel_avg_us'], d['ms_per_step']*1e3))"
done
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$o/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_avg_us'], r['frac_real'], d['extra']['cfg4_nt_100x1M_one_gpu'].get('roofline'), {k:v for k,v in d['extra']['call_latency']['spr_500x100k'].items() if 'hbm' in k or 'frac' in k}, d['cpu_baseline']['value'])"
