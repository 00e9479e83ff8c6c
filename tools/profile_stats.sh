#!/bin/bash
# tools/profile_stats.sh <tag> : the short form of profile_round.sh -- kernel-trace stats of the two bench commands, the HBM
# byte counters of their traversal kernels in separate counter-only passes, and the bench lines of the same box.
tag=${1:-r01d}
out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg2 -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/stats_cfg2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg3 -- python /root/repo/bench.py --workload cfg3_aa_200x10k --steps 10 --warmup 3 --no-cpu-baseline > $out/stats_cfg3.log 2>&1
for wl in cfg2_nt_100x50k cfg3_aa_200x10k; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${wl}_$c -- python /root/repo/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline > $out/pmc_${wl}_$c.log 2>&1 || echo "pass $wl $c failed/timeout"
  done
done
python /root/repo/bench.py --no-cpu-baseline > $out/bench_cfg2.json 2>/dev/null
python /root/repo/bench.py --no-cpu-baseline --workload cfg3_aa_200x10k > $out/bench_cfg3.json 2>/dev/null
python3 - <<PY
import csv,glob,collections,json
out='$out'
for wl in ('cfg2_nt_100x50k','cfg3_aa_200x10k'):
    acc=collections.defaultdict(list)
    for f in glob.glob(f'{out}/pmc_{wl}_*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'traverse' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    res={k:sum(v)/len(v) for k,v in sorted(acc.items())}
    json.dump(res,open(f'{out}/pmc_{wl}.json','w'),indent=1)
    print(wl,res)
for d in ('stats_cfg2','stats_cfg3'):
    for f in glob.glob(f'{out}/{d}/*/*kernel_stats.csv'):
        print(d); print(open(f).read()[:900])
for b in ('bench_cfg2','bench_cfg3'):
    print(open(f'{out}/{b}.json').read()[:700])
PY
