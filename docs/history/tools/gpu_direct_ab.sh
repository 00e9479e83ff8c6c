#!/bin/bash
# resident evaluators: how many workgroups may poll the host themselves before workgroup 0 relays (PHYHIP_RESIDENT_DIRECT)
export TMPDIR=/tmp
for rep in 1 2; do
for d in 8 16 32 1; do
  echo -n "direct<=$d: "
  a=$(PHYHIP_RESIDENT_DIRECT=$d timeout 120 python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1 | sed 's/.*record, //')
  b=$(PHYHIP_RESIDENT_DIRECT=$d timeout 120 python tools/bench_spr.py --taxa 54 --patterns 382 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['us_per_candidate'],2))")
  c=$(PHYHIP_RESIDENT_DIRECT=$d timeout 120 python tools/bench_spr.py --taxa 54 --patterns 900 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['us_per_candidate'],2))")
  e=$(PHYHIP_RESIDENT_DIRECT=$d timeout 120 python tools/bench_dlk.py 900 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['us_dLk'],2))")
  echo "trace $a | spr382 $b | spr900 $c | dlk900 $e"
done
done
