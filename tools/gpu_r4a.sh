#!/bin/bash
# round 4, first GPU call: the large-grid resident evaluator -- parity first, then numbers (one box)
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 120 phyml_amd/lib/membench 2 15 sweep > $O/membench.json 2> $O/membench.err; echo "membench rc=$?"; head -c 600 $O/membench.json; echo
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q -k "large" > $O/t_big.log 2>&1; echo "big tests rc=$?"; tail -15 $O/t_big.log
timeout 600 python -m pytest tests/test_gpu_resident.py -x -q -k "not large" > $O/t_small.log 2>&1; echo "small resident tests rc=$?"; tail -3 $O/t_small.log
timeout 600 python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_replay.py tests/test_gpu_trace.py tests/test_gpu_parity.py -x -q > $O/t_more.log 2>&1; echo "cfg5/replay/trace/parity rc=$?"; tail -3 $O/t_more.log
timeout 900 python tools/bench_big.py > $O/bench_big.jsonl 2> $O/bench_big.err; echo "bench_big rc=$?"; cat $O/bench_big.jsonl
timeout 600 python tools/bench_big.py --patterns 20000 --taxa 100 --configs launch,host_sum,device_sum > $O/bench_big_20k.jsonl 2>> $O/bench_big.err; cat $O/bench_big_20k.jsonl
