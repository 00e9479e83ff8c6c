#!/bin/bash
# round 4: PhyML's real SPR search through the drop-in boundary at sizes the large-grid resident evaluator serves --
# device-built matrices, resident against a launch per call (same box)
export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
for sz in "80 4000" "100 20000" "150 20000"; do
  for res in 0 1; do
    echo -n "device_pmat=1 PHYHIP_RESIDENT=$res $sz: "
    GLUE_DEVICE_PMAT=1 PHYHIP_RESIDENT=$res PHYHIP_RESIDENT_STATS=1 timeout 900 python tools/search_bench.py $sz --skip-host 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['device']; print({k:d[k] for k in d if k in ('seconds','lnL_final','calls')})"
  done
done
echo -n "host matrices (bit-exact route) PHYHIP_RESIDENT=1 100 20000: "
GLUE_DEVICE_PMAT=0 timeout 900 python tools/search_bench.py 100 20000 --skip-host 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['device']; print({k:d[k] for k in d if k in ('seconds','lnL_final','calls')})"
