#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -q > $O/t_res.log 2>&1; echo "resident tests rc=$?"; tail -4 $O/t_res.log
timeout 600 python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_replay.py tests/test_gpu_trace.py -x -q > $O/t_more.log 2>&1; echo "cfg5/replay/trace rc=$?"; tail -2 $O/t_more.log
for end in spr dlk eig; do
PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_big.py --label stats_$end --end $end > $O/stats_$end.log 2>&1
echo "== last command $end"; grep -E "resident|big" $O/stats_$end.log | cut -c1-330
done
timeout 900 python tools/bench_big.py --configs launch,product > $O/bench_big.jsonl 2> $O/bench_big.err; echo "bench_big rc=$?"; cut -c1-420 $O/bench_big.jsonl
timeout 600 python tools/bench_big.py --patterns 20000 --taxa 100 --configs launch,product > $O/bench_big_20k.jsonl 2>> $O/bench_big.err; cut -c1-420 $O/bench_big_20k.jsonl
timeout 600 python tools/bench_big.py --patterns 4000 --taxa 80 --configs launch,product > $O/bench_big_4k.jsonl 2>> $O/bench_big.err; cut -c1-420 $O/bench_big_4k.jsonl
