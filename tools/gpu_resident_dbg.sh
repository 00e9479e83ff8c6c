#!/bin/bash
export TMPDIR=/tmp
export PHYHIP_RESIDENT_STATS=1
for idle in 400 5000; do
echo "idle_us=$idle"
PHYHIP_RESIDENT_IDLE_US=$idle timeout 120 python tools/bench_dlk.py 382 2>&1 | tail -2 | cut -c1-300
PHYHIP_RESIDENT_IDLE_US=$idle timeout 120 python tools/bench_trace.py trace_nucleic_spr device 2>&1 | tail -2
done
