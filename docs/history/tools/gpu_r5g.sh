#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r5g; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_virtual.py -q > $o/tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $o/tests.log | tail -2; grep -E "^FAILED|^ERROR|^E  " $o/tests.log | head -30
