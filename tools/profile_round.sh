#!/bin/bash
# tools/profile_round.sh <tag> : the round's evidence run on the GPU box (gpurun): `rocprofv3 --kernel-trace --stats` of the
# default bench command (cfg2, which also runs the cfg3 line under `extra`), then separate counter-only passes (each
# bounded by `timeout`), summaries under gpurun_out/<tag>/ AND, with the hash of the kernel sources they were taken on,
# under profiles/<tag>_* (bench.py reports `roofline.traffic` only from a profile whose hash matches the sources it runs).
tag=${1:-r03}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -- python $repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-call-latency > $out/stats_default.log 2>&1
for wl in cfg2_nt_100x50k cfg3_aa_200x10k; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr " " "_" | cut -c1-40)
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${wl}_$n -- python $repo/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-companion > $out/pmc_${wl}_$n.log 2>&1 || echo "pass $wl $c failed/timeout"
  done
done
python3 - <<PY
import csv,glob,collections,json,os,sys,shutil
sys.path.insert(0,'$repo')
import bench
out='$out'; tag='$tag'; prof=os.path.join('$repo','gpurun_out',tag,'profiles'); os.makedirs(prof,exist_ok=True)
for wl in ('cfg2_nt_100x50k','cfg3_aa_200x10k'):
    acc=collections.defaultdict(list)
    for f in glob.glob(f'{out}/pmc_{wl}_*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'traverse' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    res={k:sum(v)/len(v) for k,v in sorted(acc.items())}
    res['kernel_source_hash']=bench.kernel_source_hash()
    res['hbm_bytes_per_launch']=(2.0*res.get('FETCH_SIZE',0)+res.get('WRITE_SIZE',0))*1024.0
    json.dump(res,open(f'{prof}/{tag}_pmc_{wl}.json','w'),indent=1)
    json.dump(res,open(os.path.join('$repo','profiles',f'{tag}_pmc_{wl}.json'),'w'),indent=1)  # (so that the bench line below carries roofline.traffic / frac_real)
    print(wl,res)
for f in glob.glob(f'{out}/stats_default/*/*kernel_stats.csv'):
    if 'traverse' in open(f).read():  # (the membench child process writes a stats file of its own)
        shutil.copy(f,f'{prof}/{tag}_stats_default_kernel_stats.csv'); print(open(f).read()[:1800])
PY
python $repo/bench.py --no-cpu-baseline > $out/bench_default.json 2>/dev/null
cp $out/bench_default.json $out/profiles/${tag}_bench_default.json
