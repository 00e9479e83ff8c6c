#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_cases.py tests/test_gpu_mixture.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
export PHYHIP_RESIDENT_STATS=1
for rep in 1 2; do
for r in 1 0; do
  echo "== resident=$r"
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_trace.py trace_nucleic_spr device 2>&1 | tail -1
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_trace.py trace_proteic_spr device 2>&1 | tail -2
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 382 2>&1 | tail -1 | cut -c1-160
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 1000 2>&1 | tail -1 | cut -c1-160
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 2000 aa 2>&1 | tail -1 | cut -c1-160
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 4000 2>&1 | tail -1 | cut -c1-160
done
done
