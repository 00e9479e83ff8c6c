#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_cases.py tests/test_gpu_resident.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|^E " | tail -8
export PHYHIP_RESIDENT_STATS=1
for rep in 1 2; do
for r in 1 0; do
  echo "== resident=$r"
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_trace.py trace_nucleic_spr device 2>&1 | tail -3
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_spr.py --taxa 54 --patterns 382 2>&1 | tail -2 | cut -c1-230
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_spr.py --taxa 54 --patterns 1500 2>&1 | tail -2 | cut -c1-230
done
done
