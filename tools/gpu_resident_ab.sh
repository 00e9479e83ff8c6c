#!/bin/bash
# Resident evaluator (PHYHIP_RESIDENT=1, default) against a launch per dLk (=0): the whole GPU suite, then latency, same box.
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_resident.log 2>&1
grep -E "passed|failed|Error" gpurun_out/tests_resident.log | tail -3
export PHYHIP_RESIDENT_STATS=1
for rep in 1 2; do
for r in 1 0; do
  echo "== resident=$r"
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_trace.py trace_nucleic_spr device 2>&1 | tail -2
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_trace.py trace_proteic_spr device 2>&1 | tail -2
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 382 2>&1 | tail -1 | cut -c1-200
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 2000 aa 2>&1 | tail -1 | cut -c1-200
  PHYHIP_RESIDENT=$r timeout 120 python tools/bench_dlk.py 4000 2>&1 | tail -1 | cut -c1-200
done
done
