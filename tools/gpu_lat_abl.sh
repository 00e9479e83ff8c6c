#!/bin/bash
# Latency per call of several builds (phyml_amd/<dir>) on the same box, interleaved and repeated.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in "$@"; do
  a=$(PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1 | sed 's/.*= //')
  b=$(PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_nucleic_spr 2>/dev/null | tail -1 | sed 's/.*record, //')
  c=$(PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['us_per_candidate'],2))")
  d=$(PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py --candidates 1000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['us_per_candidate'],2))")
  echo "$lib | dev-pmat: $a | host-pmat: $b | spr382: $c | cfg5: $d"
done
done
