#!/bin/bash
# round 4: the large-grid resident evaluator -- parity, then numbers, then where a command's time goes (one box)
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/t_res.log 2>&1; echo "resident tests rc=$?"; tail -5 $O/t_res.log
timeout 600 python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_replay.py tests/test_gpu_trace.py -x -q > $O/t_more.log 2>&1; echo "cfg5/replay/trace rc=$?"; tail -3 $O/t_more.log
timeout 900 python tools/bench_big.py > $O/bench_big.jsonl 2> $O/bench_big.err; echo "bench_big rc=$?"; cat $O/bench_big.jsonl
timeout 600 python tools/bench_big.py --patterns 20000 --taxa 100 --configs launch,host_sum,device_sum > $O/bench_big_20k.jsonl 2>> $O/bench_big.err; cat $O/bench_big_20k.jsonl
for mode in 100000000 0; do
PHYHIP_RESIDENT_STATS=1 PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_BIG_DEVICE_SUM=$mode timeout 300 python tools/bench_big.py --label stats_$mode > $O/stats_$mode.log 2>&1; grep -E "resident|big" $O/stats_$mode.log | cut -c1-250
done
