#!/bin/bash
# timing-only ablations of the 20-state kernel (results invalid under PHYHIP_ABLATE / PHYHIP_NOLOADS)
for p in 4096 10000; do
  for cfg in "0 0" "0 1" "1 0" "2 0" "4 0" "4 1" "5 1" "7 1"; do
    set -- $cfg
    r=$(PHYHIP_ABLATE=$1 PHYHIP_NOLOADS=$2 timeout 120 python bench.py --workload cfg3_aa_200x10k --patterns $p --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_us'])")
    echo "P=$p ablate=$1 noloads=$2 kernel_us=$r"
  done
done
