#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_trace.py tests/test_gpu_resident.py -q > $O/t_k.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_k.log; grep -E "^FAILED|^ERROR|^E  " $O/t_k.log | head
