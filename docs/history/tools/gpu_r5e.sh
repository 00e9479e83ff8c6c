#!/bin/bash
# round 5: in-step tip x tip children in the 20-state kernel -- parity first, then cfg3 with / without
export TMPDIR=/tmp
o=gpurun_out/r5e; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_virtual.py tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_fuzz.py tests/test_gpu_trace.py tests/test_gpu_mixture.py -q -x > $o/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $o/tests.log | tail -2; grep -E "^FAILED|^ERROR" $o/tests.log | head; tail -25 $o/tests.log | head -40
for inl in 1 0; do
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_VIRT_INLINE=$inl timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $o/bench_cfg3_inl$inl.json 2> $o/bench_cfg3_inl$inl.err
  python -c "
import json; d=json.load(open('$o/bench_cfg3_inl$inl.json')); r=d['roofline']; print('cfg3 in-step $inl ms/step %.4f kernel %.1f frac %.3f stored %s lnLerr %s vb %s' % (d['ms_per_step'], r['kernel_avg_us'], r['frac'], r.get('all_buffers_stored',{}).get('kernel_avg_us'), d.get('lnL_rel_err'), r['virtual_buffers']))"
done
timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $o/bench_aa100k.json 2> $o/bench_aa100k.err
python -c "
import json; d=json.load(open('$o/bench_aa100k.json')); r=d['roofline']; print('aa 100k ms/step', d['ms_per_step'], 'kernel us', r['kernel_avg_us'], 'frac', r['frac'], r.get('all_buffers_stored'))"
