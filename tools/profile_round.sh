#!/bin/bash
# tools/profile_round.sh <tag> : the round's evidence run on the GPU box (gpurun): kernel-trace stats for the
# default bench command, then separate counter-only passes (each bounded by `timeout`), results under gpurun_out/<tag>/.
tag=${1:-r01}
out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg2 -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/stats_cfg2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg3 -- python /root/repo/bench.py --workload cfg3_aa_200x10k --steps 10 --warmup 3 --no-cpu-baseline > $out/stats_cfg3.log 2>&1
for wl in cfg2_nt_100x50k cfg3_aa_200x10k; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA"; do
    n=$(echo $c | tr " " "_" | cut -c1-40)
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${wl}_$n -- python /root/repo/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline > $out/pmc_${wl}_$n.log 2>&1 || echo "pass $wl $c failed/timeout"
  done
done
python3 - <<PY
import csv,glob,collections,json,os
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
        print(d); print(open(f).read()[:1500])
PY
