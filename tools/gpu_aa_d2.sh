#!/bin/bash
# 20-state kernel's list form: loads two operations ahead (default where workgroups are small) against one (diag: PHYHIP_AA_D2=0), one
# lease: the 20-state parity tests first, then cfg3 and 2 500 / 25 000 patterns
export TMPDIR=/tmp
o=gpurun_out/aa_d2; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_virtual.py tests/test_gpu_fuzz.py tests/test_gpu_mixture.py -q -x -k "aa or 20 or proteic or lg4x or mixture or fuzz or virtual" > $o/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $o/tests.log
for rep in 1 2 3; do
for d2 in 1 0; do
  for pat in "" "--patterns 5000" "--patterns 25000"; do
    PHYHIP_LIBDIR=phyml_amd/lib_diag PHYHIP_AA_D2=$d2 timeout 300 python bench.py --workload cfg3_aa_200x10k $pat --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $o/b.json 2> $o/b.err || tail -3 $o/b.err
    python - "$d2" "$pat" <<'P'
import json,sys
d=json.load(open('bench_detail.json')); r=d['roofline']
print('D2',sys.argv[1],sys.argv[2] or 'cfg3', 'kernel %.1f us step %.1f us'%(r['kernel_avg_us'], d['ms_per_step']*1e3), r['kernel'], 'stored %.1f'%r.get('all_buffers_stored',{}).get('kernel_avg_us',0), 'lnL_err', d.get('lnL_rel_err'))
P
  done
done
done
