#!/bin/bash
# round 5: pmat20_kernel gives the bits of pmat_kernel<20> (the switch tests that touch it, diag build)
export TMPDIR=/tmp; o=gpurun_out/r5s; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_switches.py -q -k "PMAT or default or GENERIC_AA or AA_NW" > $o/switches.log 2>&1; echo "switches rc=$?"; grep -E "passed|failed" $o/switches.log | tail -1; grep -E "^FAILED|^E  " $o/switches.log | head -10
