#!/bin/bash
# tools/pmc2.sh <tag> [bench args...] : compact PMC set (separate counter-only passes) for the traversal kernel
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
out=/root/repo/gpurun_out/pmc_$tag; mkdir -p $out
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
)
i=0
for c in "${passes[@]}"; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(list)
dur=[]
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'traverse' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$out/p3/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'traverse' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
res={k:sum(v)/len(v) for k,v in sorted(acc.items())}
if dur: res['kernel_us_in_GRBM_pass']=sum(dur)/len(dur)
json.dump(res,open('$out/summary.json','w'),indent=1)
for k in sorted(res): print(f"{k:40s} {res[k]:18.1f}")
PY
