#!/bin/bash
# lane-per-pattern nucleotide kernel: loads two operations ahead (194 VGPRs, 2 waves per SIMD at G = 2) against one ahead
# (156 VGPRs, 3 waves per SIMD); parity first, then cfg2 / 125 k / 1 M on the same box
export TMPDIR=/tmp
PHYHIP_NT2_DIST=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cases.py tests/test_gpu_replay.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do
for d in 2 1; do
  for g in "" 1; do
  echo -n "dist=$d groups=${g:-auto}: "
  for P in 50000 125000 1000000; do
    PHYHIP_NT_GROUPS=$g PHYHIP_NT2_DIST=$d timeout 200 python bench.py --patterns $P --steps 40 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$P step %.1f kernel %.1f err %s |' % (d['ms_per_step']*1e3, d['roofline']['kernel_avg_us'], d.get('lnL_rel_err')), end=' ')"
  done; echo
  done
done
done
