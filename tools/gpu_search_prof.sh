#!/bin/bash
# kernel-time share of PhyML's real SPR search on the device (the driver process itself under rocprofv3 --stats)
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/search_prof; rm -rf $out; mkdir -p $out
cd /tmp
SEARCH_BENCH_WRAP="rocprofv3 --kernel-trace --stats --output-format csv -d $out --" timeout 900 python $GRAFT_REPO_ROOT/tools/search_bench.py ${1:-80} ${2:-4000} --skip-host > $out/log.txt 2>&1
for f in $out/*/*kernel_stats.csv; do echo $f; head -12 $f | cut -c1-220; done
tail -1 $out/log.txt | cut -c1-200
