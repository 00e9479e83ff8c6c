#!/bin/bash
# round 3: cfg5 SPR candidate and the small-alignment replay with host-computed matrices in the kernel arguments
# (PHYHIP_ARG_UPLOADS) against the upload kernel, and against device-built matrices -- one box
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
for rep in 1 2; do
  timeout 300 python tools/bench_spr.py 2>/dev/null | tail -1 | cut -c1-330
  PHYHIP_ARG_UPLOADS=1 timeout 300 python tools/bench_spr.py --host-pmat 2>/dev/null | tail -1 | cut -c1-330
  PHYHIP_ARG_UPLOADS=0 timeout 300 python tools/bench_spr.py --host-pmat 2>/dev/null | tail -1 | cut -c1-330
done
for rep in 1 2; do
  timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 5000 2>/dev/null | tail -1 | cut -c1-330
  PHYHIP_ARG_UPLOADS=1 timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 5000 --host-pmat 2>/dev/null | tail -1 | cut -c1-330
  PHYHIP_ARG_UPLOADS=0 timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 5000 --host-pmat 2>/dev/null | tail -1 | cut -c1-330
done
