#!/bin/bash
# round 3: PhyML's real SPR search through the drop-in boundary (device-driven), nucleotides and amino acids, both matrix routes
export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
for dp in 0 1; do
  for sz in "54 382" "80 4000" "40 1500 --aa" "60 6000 --aa"; do
    echo -n "device_pmat=$dp $sz: "
    GLUE_DEVICE_PMAT=$dp timeout 400 python tools/search_bench.py $sz --skip-host 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['device']; print({k:d[k] for k in d if k in ('seconds','lnL_final','calls')})"
  done
done
