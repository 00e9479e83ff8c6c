#!/bin/bash
# round 4: where a command of the large-grid resident evaluator spends its time (stamps of the last command, by kind)
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -q > $O/t_res.log 2>&1; echo "resident tests rc=$?"; tail -8 $O/t_res.log
PHYHIP_RESIDENT_DEBUG=1 PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag timeout 300 python -m pytest tests/test_gpu_resident.py -x -q -k "large_grid_resident_evaluator_spr and 6000" > $O/t_c1.log 2>&1; grep -E "^big" $O/t_c1.log | sed -n '1,12p;60,72p' | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_replay.py tests/test_gpu_trace.py tests/test_gpu_parity.py tests/test_gpu_cases.py -x -q > $O/t_more.log 2>&1; echo "cfg5/replay/trace/parity/cases rc=$?"; tail -3 $O/t_more.log
for mode in 100000000 0; do for end in spr dlk eig; do
PHYHIP_RESIDENT_STATS=1 PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_BIG_DEVICE_SUM=$mode timeout 300 python tools/bench_big.py --label stats_${mode}_$end --end $end > $O/stats_${mode}_$end.log 2>&1
echo "== sum threshold $mode, last command $end"; grep -E "resident|big" $O/stats_${mode}_$end.log | cut -c1-330
done; done
timeout 900 python tools/bench_big.py --configs launch,product > $O/bench_big.jsonl 2> $O/bench_big.err; echo "bench_big rc=$?"; cut -c1-420 $O/bench_big.jsonl
timeout 600 python tools/bench_big.py --patterns 20000 --taxa 100 --configs launch,host_sum,device_sum > $O/bench_big_20k.jsonl 2>> $O/bench_big.err; cut -c1-420 $O/bench_big_20k.jsonl
