#!/bin/bash
# tools/profile_stats.sh <tag>: only the `rocprofv3 --kernel-trace --stats` pass of tools/profile_round.sh (the default bench
# without its call_latency part, whose launches share kernel names with the bench's) -> gpurun_out/<tag>/profiles/
tag=${1:-r02}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag; mkdir -p $out/profiles
cd /tmp; export TMPDIR=/tmp
rm -rf $out/stats_default
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -- python $repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-call-latency > $out/stats_default.log 2>&1
for f in $out/stats_default/*/*kernel_stats.csv; do
  if grep -q traverse $f; then cp $f $out/profiles/${tag}_stats_default_kernel_stats.csv; head -6 $f | cut -c1-70,190-290; fi
done
grep '^{"metric"' $out/stats_default.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'kernel us', d['roofline']['kernel_avg_us'], 'traffic', d['roofline']['traffic'], 'cfg3 kernel us', d['extra']['cfg3_aa_200x10k']['roofline']['kernel_avg_us'])"
