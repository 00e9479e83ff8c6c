#!/bin/bash
# dlk_kernel grid size A/B (diag build): wall microseconds per dLk at several sizes
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
export PHYHIP_LIBDIR=$repo/phyml_amd/lib_diag
for args in "100000" "100000 aa" "1000000" "20000" "20000 aa"; do
  for g in 512 1024 2048 4096; do
    echo -n "P=$args grid $g: "; PHYHIP_DLK_GRID=$g timeout 200 python tools/bench_dlk.py $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dLk', round(d['us_dLk'],1), 'us; Lk_edge', round(d['us_Lk_edge'],1), 'eigen', round(d['us_Update_Eigen_Lr'],1))"
  done
done
