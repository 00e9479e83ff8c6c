#!/bin/bash
# round 5: consumer waves per workgroup of the 20-state kernel on a SMALL alignment's SPR candidates (diag build, PHYHIP_AA_NW)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r5t; mkdir -p $o
export PHYHIP_LIBDIR=$R/phyml_amd/lib_diag
for nw in 1 3 7; do
  echo "AA_NW=$nw: $(PHYHIP_AA_NW=$nw python $R/tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["us_per_candidate"])')"
done
