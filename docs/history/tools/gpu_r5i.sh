#!/bin/bash
# round 5, final evidence: the whole GPU suite, the round's profiles (kernel stats + counter passes of cfg2 / cfg3 + the bench line
# with counter traffic), the counter profiles of the other kernels, smoke
export TMPDIR=/tmp
o=gpurun_out/r5i; mkdir -p $o
timeout 1800 python -m pytest tests -q -m gpu > $o/full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 bash tools/profile_round.sh r05 > $o/profile_round.log 2>&1; tail -30 $o/profile_round.log | cut -c1-400
timeout 900 bash tools/profile_r05_extra.sh r05 > $o/profile_extra.log 2>&1; grep -E "resident_big|eigen_lr|traverse_nt2|pmat" $o/profile_extra.log | cut -c1-330
