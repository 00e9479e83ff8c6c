#!/bin/bash
# round 4: the final sum through one partial sum per workgroup -- parity first, then the three forms side by side
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_cfg5.py "tests/test_gpu_cases.py::test_alias_subpatt_gate_is_mirrored" -q -x > $O/t_l.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_l.log; grep -E "^FAILED|^ERROR|^E  " $O/t_l.log | head
timeout 600 python tools/bench_big.py --configs product,tickets,diag,product,tickets > $O/big_l.jsonl 2> $O/big_l.err; cat $O/big_l.jsonl
PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_big.py --configs product > $O/big_l_stats.jsonl 2> $O/big_l_stats.err; tail -25 $O/big_l_stats.err
