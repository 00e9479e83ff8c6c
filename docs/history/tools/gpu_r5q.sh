#!/bin/bash
# round 5: where a 20-state SPR candidate's time goes (two launches per candidate): kernel durations under rocprofv3 next to the unprofiled call time
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r5q; mkdir -p $o
for shape in "37 429" "200 10000"; do
  set -- $shape
  python $R/tools/bench_spr.py --taxa $1 --patterns $2 --states 20 --candidates 3000 2>&1 | tail -1 | tee $o/plain_$1x$2.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$1x$2 -o aa -- python $R/tools/bench_spr.py --taxa $1 --patterns $2 --states 20 --candidates 3000 > $o/prof_$1x$2.log 2>&1
  f=$(find $o/prof_$1x$2 -name "*kernel_stats.csv" | head -1); echo "== $shape: $f"; head -8 "$f" | cut -c1-260
  find $o/prof_$1x$2 -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $o/prof_$1x$2 -name "*.db" -delete
done
