#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_prof; rm -rf $out; mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_trace.py trace_nucleic_spr device > $out/log.txt 2>&1
for f in $out/*/*kernel_stats.csv; do head -8 $f | cut -c1-260; done
tail -1 $out/log.txt
python $GRAFT_REPO_ROOT/tools/bench_trace.py trace_nucleic_spr device | tail -1
PHYHIP_SPIN=0 python $GRAFT_REPO_ROOT/tools/bench_trace.py trace_nucleic_spr device | tail -1
