#!/bin/bash
export TMPDIR=/tmp
export PHYHIP_RESIDENT_STATS=1
timeout 60 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 300 2>&1 | tail -2 | cut -c1-230
