#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_cfg5.py tests/test_gpu_cases.py -q > $O/t_res.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_res.log; grep -E "^FAILED" $O/t_res.log
timeout 900 python tools/bench_big.py --configs launch,product > $O/bench_big.jsonl 2> $O/bench_big.err; echo "bench_big rc=$?"; cut -c1-330 $O/bench_big.jsonl
timeout 600 python tools/bench_big.py --patterns 20000 --taxa 100 --configs launch,product > $O/bench_big_20k.jsonl 2>> $O/bench_big.err; cut -c1-330 $O/bench_big_20k.jsonl
timeout 600 python tools/bench_big.py --patterns 4000 --taxa 80 --configs launch,product > $O/bench_big_4k.jsonl 2>> $O/bench_big.err; cut -c1-330 $O/bench_big_4k.jsonl
PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_big.py --label stats_spr --end spr > $O/stats_spr.log 2>&1; grep -E "resident|big" $O/stats_spr.log | cut -c1-200
