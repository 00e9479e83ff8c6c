#!/bin/bash
# round 5: counter profiles of the kernels that had none, the shim's host share in a real search, and the two tests touched since r5c
export TMPDIR=/tmp
o=gpurun_out/r5d; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_virtual.py -q > $o/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $o/tests.log | tail -2; grep -E "^FAILED|^ERROR" $o/tests.log | head
timeout 900 bash tools/profile_r05_extra.sh r05x > $o/profile_extra.log 2>&1; tail -40 $o/profile_extra.log
timeout 400 python tools/host_share.py 150 20000 > $o/host_share_150x20000.txt 2>&1; tail -14 $o/host_share_150x20000.txt
timeout 300 python tools/host_share.py 150 20000 --device-pmat > $o/host_share_150x20000_device_pmat.txt 2>&1; tail -14 $o/host_share_150x20000_device_pmat.txt
timeout 200 python tools/host_share.py 54 382 > $o/host_share_54x382.txt 2>&1; tail -14 $o/host_share_54x382.txt
