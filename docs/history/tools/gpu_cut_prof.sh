#!/bin/bash
# Timing attribution of the short-launch traversal kernel: builds with parts cut out (results invalid), rocprof averages
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in "$@"; do
  out=$R/gpurun_out/cut_prof/$lib; rm -rf $out; mkdir -p $out
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/tools/bench_trace.py trace_nucleic_spr device > $out/log.txt 2>&1
  echo "== $lib: $(tail -1 $out/log.txt | sed 's/.*record, //')"
  for f in $out/*/*kernel_stats.csv; do grep -E "true>|dlk_kernel" $f | sed 's/(phyhip::[^"]*"/"/' | cut -d, -f1-4,6-7; done
done
