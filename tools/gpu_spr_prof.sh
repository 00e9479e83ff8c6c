#!/bin/bash
# kernel durations of the cfg5 SPR-candidate stream (rocprofv3 --kernel-trace --stats), matrices rebuilt inside the traversal launch or not
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for d in 1 0; do
  out=$repo/gpurun_out/r03_spr_prof_dist$d; rm -rf $out; mkdir -p $out
  PHYHIP_DIST_PMAT=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $repo/tools/bench_spr.py --candidates 1500 > $out/log.txt 2>&1
  tail -1 $out/log.txt | cut -c1-200
  cat $out/*/*kernel_stats.csv | cut -c1-220 | head -8
done
