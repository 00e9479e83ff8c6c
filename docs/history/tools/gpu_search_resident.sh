#!/bin/bash
# PhyML's real SPR search through the drop-in boundary (device-driven), resident evaluators on / off, same box
export TMPDIR=/tmp
export PHYHIP_RESIDENT_STATS=1
for dp in 0 1; do
for r in 1 0 1 0; do
  for sz in "54 382" "80 1500"; do
    echo -n "device_pmat=$dp resident=$r $sz: "
    GLUE_DEVICE_PMAT=$dp PHYHIP_RESIDENT=$r timeout 300 python tools/search_bench.py $sz --skip-host 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['device']; print({k:d[k] for k in d if k in ('seconds','lnL')})"
  done
done
done
