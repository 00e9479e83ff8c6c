#!/bin/bash
# Host-side final sum (posted block records) against the in-kernel ticket / final_reduce paths, same box:
# parity tests first, then latency per scalar-returning call at several sizes.
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_mixture.py tests/test_gpu_cases.py -x -q -m gpu > gpurun_out/tests_hostsum.log 2>&1
grep -E "passed|failed|Error" gpurun_out/tests_hostsum.log | tail -3
for hs in 1 0 1 0; do
  echo "host_sum=$hs"
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_dlk.py 382 2>/dev/null | tail -1
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_dlk.py 50000 2>/dev/null | tail -1
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_dlk.py 4000 aa 2>/dev/null | tail -1
done
for hs in 1 0; do
  echo "host_sum=$hs"
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_spr.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg5 us/candidate', round(d['us_per_candidate'],2), 'full ms', round(d['full_both_sides_Lk_ms'],2))"
  PHYHIP_HOST_SUM=$hs timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg2 step_us', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"
  PHYHIP_HOST_SUM=$hs timeout 300 python tools/bench_mixture.py 2>/dev/null | tail -2
done
