#!/bin/bash
# round 4: command records pushed into device memory -- parity under each allocation kind, then timing
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag
for k in 2 1 3; do
  PHYHIP_PUSH_CMDS=$k timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_cfg5.py -q -x > $O/t_p$k.log 2>&1; echo "push $k tests rc=$?"; grep -E "passed|failed" $O/t_p$k.log | tail -1; grep -E "^FAILED|^ERROR|^E  " $O/t_p$k.log | head -5
done
unset PHYHIP_LIBDIR
timeout 900 python tools/bench_big.py --configs diag,push1,push2,push3,diag,push2 > $O/big_p.jsonl 2> $O/big_p.err; cut -c1-250 $O/big_p.jsonl
PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag timeout 300 python tools/bench_trace.py > $O/trace_p0.txt 2>&1; tail -4 $O/trace_p0.txt | cut -c1-300
PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_PUSH_CMDS=2 timeout 300 python tools/bench_trace.py > $O/trace_p2.txt 2>&1; tail -4 $O/trace_p2.txt | cut -c1-300
