#!/bin/bash
# round 5: in-step tip x tip children (lane-per-pattern kernel) -- the whole GPU suite, then the traversal kernels with / without
export TMPDIR=/tmp
o=gpurun_out/r5c; mkdir -p $o
timeout 1800 python -m pytest tests -q -m gpu > $o/full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -20
for inl in 1 0; do
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_VIRT_INLINE=$inl timeout 300 python bench.py --workload cfg2_nt_100x50k --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $o/bench_cfg2_inl$inl.json 2> $o/bench_cfg2_inl$inl.err
  python -c "
import json; d=json.load(open('$o/bench_cfg2_inl$inl.json')); r=d['roofline']; print('cfg2 in-step $inl ms/step %.4f kernel %.1f stored %s lnLerr %s vb %s' % (d['ms_per_step'], r['kernel_avg_us'], r.get('all_buffers_stored',{}).get('kernel_avg_us'), d.get('lnL_rel_err'), r['virtual_buffers']))"
done
for wl in cfg2_nt_100x50k cfg3_aa_200x10k cfg4_nt_100x1M cfg4_nt_100x125k; do
timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $o/bench_$wl.json 2> $o/bench_$wl.err
python -c "
import json; d=json.load(open('$o/bench_$wl.json')); r=d['roofline']; print('$wl ms/step', d['ms_per_step'], 'kernel us', r['kernel_avg_us'], 'value', d['value'], 'frac', r['frac'], 'lnLerr', d.get('lnL_rel_err'), r.get('all_buffers_stored'))"
done
