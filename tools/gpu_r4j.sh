#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_trace.py tests/test_gpu_replay.py -q > $O/t_j.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_j.log; grep -E "^FAILED|^ERROR" $O/t_j.log | head
for r in 0 1; do echo "== PHYHIP_RESIDENT=$r"; PHYHIP_RESIDENT=$r PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_trace.py trace_proteic_spr device 2>&1 | tail -4; PHYHIP_RESIDENT=$r timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>&1 | tail -2; done
timeout 300 python tools/bench_dlk.py 2000 aa 2>&1 | tail -3
PHYHIP_RESIDENT=0 timeout 300 python tools/bench_dlk.py 2000 aa 2>&1 | tail -3
