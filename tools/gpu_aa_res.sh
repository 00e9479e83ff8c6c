#!/bin/bash
# resident 20-state evaluator: SPR candidates at 37 x 429 aa, product against PHYHIP_RESIDENT=0 and builds with parts cut out
export TMPDIR=/tmp
for rep in 1 2; do
  echo "launch path:"; PHYHIP_RESIDENT=0 python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  us/candidate %.2f'%d['us_per_candidate'])"
  for lib in lib "$@"; do
    echo "$lib:"; PHYHIP_LIBDIR=phyml_amd/$lib python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  us/candidate %.2f'%d['us_per_candidate'])"
  done
done
