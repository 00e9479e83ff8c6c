#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in lib_base lib; do
  echo -n "$lib: "
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py --candidates 1500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 us/candidate', round(d['us_per_candidate'],2), 'full ms', round(d['full_both_sides_Lk_ms'],2))"
done
done
