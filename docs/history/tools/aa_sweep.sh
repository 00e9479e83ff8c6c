#!/bin/bash
# timing-only ablations of the 20-state kernels (diag build; results invalid under PHYHIP_ABLATE / PHYHIP_NOLOADS):
# ablate 2 = no cross-category exchange/barrier, 4 = A-fragment loads return nothing, noloads 1 = child loads return nothing
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag
for p in 10000 100000; do
  for cfg in "0 0" "2 0" "4 0" "6 0" "0 1" "6 1"; do
    set -- $cfg
    r=$(PHYHIP_AA_DIST=1 PHYHIP_ABLATE=$1 PHYHIP_NOLOADS=$2 timeout 120 python bench.py --workload cfg3_aa_200x10k --patterns $p --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_us'])")
    echo "P=$p ablate=$1 noloads=$2 kernel_us=$r"
  done
done
