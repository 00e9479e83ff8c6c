#!/bin/bash
# tools/timeline.sh [workload] : kernel begin/end timeline of a few bench steps (gaps between the kernels of a step)
wl=${1:-cfg2_nt_100x50k}
out=/root/repo/gpurun_out/timeline_$wl; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python /root/repo/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > $out/run.log 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob('$out/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]))
rows=[r for r in rows if any(k in r[2] for k in ('traverse','pmat','final_reduce','frag','eigen','dlk'))]
rows.sort()
prev=None
for s,e,n in rows[-24:]:
    gap = (s-prev)/1000 if prev else 0
    print(f"gap {gap:8.2f} us  dur {(e-s)/1000:8.2f} us  {n}")
    prev=e
PY
