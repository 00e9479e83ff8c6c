#!/bin/bash
# 20-state kernel, one against two wave-tiles per consumer wave (diag build: PHYHIP_AA_NT), one lease: parity tests first, then cfg3
# and 100 000 patterns, kernel time by HIP events (bench.py --workload cfg3_aa_200x10k)
export TMPDIR=/tmp
o=gpurun_out/aa_nt; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_virtual.py tests/test_gpu_fuzz.py tests/test_gpu_mixture.py -q -x -k "aa or 20 or proteic or lg4x or mixture or fuzz or virtual" > $o/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $o/tests.log
for rep in 1 2; do
for nt in 1 2; do
  for pat in "" "--patterns 100000"; do
    PHYHIP_LIBDIR=phyml_amd/lib_diag PHYHIP_AA_NT=$nt timeout 300 python bench.py --workload cfg3_aa_200x10k $pat --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $o/b.json 2> $o/b.err || tail -3 $o/b.err
    python - "$nt" "$pat" <<'P'
import json,sys
d=json.load(open('bench_detail.json')); r=d['roofline']
print('NT',sys.argv[1],sys.argv[2] or 'cfg3', 'kernel %.1f us step %.1f us'%(r['kernel_avg_us'], d['ms_per_step']*1e3), r['kernel'], 'stored %.1f'%r.get('all_buffers_stored',{}).get('kernel_avg_us',0), 'lnL_err', d.get('lnL_rel_err'))
P
  done
done
done
