#!/bin/bash
# round 4: the large-grid kernel launched for one evaluation (one-shot) -- parity, then launch / launch_old / product side by side
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_cfg5.py tests/test_gpu_switches.py tests/test_gpu_trace.py -q -x > $O/t_o.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/t_o.log | tail -2; grep -E "^FAILED|^ERROR|^E  " $O/t_o.log | head
timeout 900 python tools/bench_big.py --configs launch,launch_old,product,launch,launch_old > $O/big_o.jsonl 2> $O/big_o.err; cut -c1-330 $O/big_o.jsonl
timeout 600 python tools/bench_big.py --patterns 20000 --taxa 100 --configs launch,launch_old,product > $O/big_o20k.jsonl 2>> $O/big_o.err; cut -c1-330 $O/big_o20k.jsonl
