#!/bin/bash
# round 5, first contact: the virtual-buffer tests, the parity suites they touch, and the two traversal kernels with / without them
export TMPDIR=/tmp
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_virtual.py -x -q > $o/virtual.log 2>&1; echo "virtual rc=$?"; tail -15 $o/virtual.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_fuzz.py -x -q > $o/parity.log 2>&1; echo "parity rc=$?"; tail -5 $o/parity.log
for v in 16 0; do
  for wl in cfg2_nt_100x50k cfg3_aa_200x10k; do
    timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-extra --virtual-buffers $v > $o/bench_${wl}_v$v.json 2> $o/bench_${wl}_v$v.err
    python - <<PY
import json
d=json.load(open('$o/bench_${wl}_v$v.json')); r=d['roofline']
print('$wl', 'virtual', $v, 'ms/step %.4f kernel_us %.1f value %.0f frac %.3f write %.3g lnL_err %s vb %s' % (d['ms_per_step'], r['kernel_avg_us'], d['value'], r['frac'], r['write_bytes'], d.get('lnL_rel_err'), r['virtual_buffers']))
PY
  done
done
timeout 300 python bench.py --workload cfg4_nt_100x1M --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $o/bench_1M.json 2> $o/bench_1M.err
python -c "
import json; d=json.load(open('$o/bench_1M.json')); r=d['roofline']; print('1M ms/step', d['ms_per_step'], 'kernel us', r['kernel_avg_us'], 'value', d['value'], r.get('all_buffers_stored'))"
