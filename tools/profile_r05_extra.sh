#!/bin/bash
# tools/profile_r05_extra.sh [tag]: counter profiles of the kernels that had none (round-4 verdict, weak #5): the large-grid kernel's
# one-shot form per call kind at the cfg5 size, eigen_lr_kernel, pmat_kernel, and the 1 M-pattern traversal.  Separate counter-only
# passes (--pmc with --kernel-trace only), every pass bounded; per-kernel averages -> profiles/<tag>_pmc_cfg5_<kind>.json, <tag>_pmc_cfg4_1M.json
tag=${1:-r05}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
passes=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE")
run() { # name, command...
  name=$1; shift
  i=0
  for c in "${passes[@]}"; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${name}_p$i -- "$@" > $out/pmc_${name}_p$i.log 2>&1 || echo "pass $name $i failed/timeout"
    i=$((i+1))
  done
}
for kind in spr dlk eig; do run cfg5_$kind python $repo/tools/pmc_cfg5.py $kind --n 40; done
run cfg4_1M python $repo/bench.py --workload cfg4_nt_100x1M --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-companion
python3 - <<PY
import csv,glob,collections,json,os,sys
sys.path.insert(0,'$repo')
import bench
out='$out'; tag='$tag'; prof=os.path.join(out,'profiles'); os.makedirs(prof,exist_ok=True)
def short(n):
    n=n.split('(')[0]
    return n.replace('phyhip::','').replace('void ','').strip()
for name in ('cfg5_spr','cfg5_dlk','cfg5_eig','cfg4_1M'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for f in glob.glob(f'{out}/pmc_{name}_p*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(f'{out}/pmc_{name}_p4/*/*kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    res={'kernel_source_hash':bench.kernel_source_hash(),'unit':'per dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB (fetch to be doubled on gfx950)','kernels':{}}
    for k,v in acc.items():
        d={c:sum(x)/len(x) for c,x in sorted(v.items())}
        d['dispatches']=max(len(x) for x in v.values())
        if dur.get(k): d['kernel_us_under_profiler']=sum(dur[k])/len(dur[k])
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            d['hbm_bytes_per_dispatch']=(2.0*d['FETCH_SIZE']+d['WRITE_SIZE'])*1024.0
            if d.get('kernel_us_under_profiler'): d['hbm_TBps_under_profiler']=d['hbm_bytes_per_dispatch']/d['kernel_us_under_profiler']/1e6
        res['kernels'][k]=d
    json.dump(res,open(f'{prof}/{tag}_pmc_{name}.json','w'),indent=1)
    for k,d in res['kernels'].items():
        print(name,k[:70],{a:round(b,1) for a,b in d.items() if a in ('dispatches','kernel_us_under_profiler','hbm_bytes_per_dispatch','hbm_TBps_under_profiler','SQ_WAIT_ANY','SQ_WAVE_CYCLES','SQ_WAIT_INST_ANY')})
PY
