#!/bin/bash
# round 3: stamps of one consumer wave of the second-generation 20-state kernel + counter passes (one gpurun call)
export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r03a; mkdir -p $out
cd $repo
PHYHIP_LIBDIR=$repo/phyml_amd/lib_diag PHYHIP_ABLATE=8 timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-call-latency > $out/s8.json 2>$out/stamps_cfg3.txt
head -70 $out/stamps_cfg3.txt
cd /tmp
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_LDS"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum"
)
i=0
for c in "${passes[@]}"; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -- python $repo/bench.py --workload cfg3_aa_200x10k --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-call-latency > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'traverse' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc): print(f"{k:40s} {sum(acc[k])/len(acc[k]):18.1f}  (n={len(acc[k])})")
PY
