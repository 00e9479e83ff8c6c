#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_shard.py tests/test_gpu_mixture.py tests/test_gpu_cases.py -q > $O/t_i.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_i.log; grep -E "^FAILED|^ERROR" $O/t_i.log | head
timeout 600 python tools/multigpu_selfcheck.py --devices 0,0 --patterns 200000 --steps 10 > $O/selfcheck.log 2>&1; echo "selfcheck rc=$?"; tail -22 $O/selfcheck.log
