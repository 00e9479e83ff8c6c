#!/bin/bash
# round 5: the 20-state matrix rebuild folded into short launches (phyhip_aa.hpp): switch tests (same numbers as the separate
# launch), the 20-state parity tests incl. PhyML's real proteic search, and SPR candidates with / without the fold
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r5u; mkdir -p $o
cd $R
timeout 600 python -m pytest tests/test_gpu_switches.py -q -k "default or FOLD_PMATS or PMAT or AA_NW or GENERIC_AA or ARGS_RECS or RESIDENT" > $o/switches.log 2>&1; echo "switches rc=$?"; grep -E "passed|failed" $o/switches.log | tail -1; grep -E "^FAILED|^E  " $o/switches.log | head -12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixture.py tests/test_gpu_virtual.py tests/test_gpu_replay.py tests/test_gpu_cases.py tests/test_gpu_trace.py tests/test_gpu_search.py -q > $o/subset.log 2>&1; echo "subset rc=$?"; grep -E "passed|failed" $o/subset.log | tail -1; grep -E "^FAILED|^E  " $o/subset.log | head -12
cd /tmp
for shape in "37 429" "200 10000"; do
  set -- $shape
  echo "product $1x$2: $(python $R/tools/bench_spr.py --taxa $1 --patterns $2 --states 20 --candidates 3000 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["us_per_candidate"], d["lnL"])')"
  echo "diag fold off $1x$2: $(PHYHIP_LIBDIR=$R/phyml_amd/lib_diag PHYHIP_FOLD_PMATS=0 python $R/tools/bench_spr.py --taxa $1 --patterns $2 --states 20 --candidates 3000 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["us_per_candidate"], d["lnL"])')"
  echo "diag fold on  $1x$2: $(PHYHIP_LIBDIR=$R/phyml_amd/lib_diag python $R/tools/bench_spr.py --taxa $1 --patterns $2 --states 20 --candidates 3000 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["us_per_candidate"], d["lnL"])')"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p_fold -o aa -- python $R/tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 > $o/p_fold.log 2>&1
f=$(find $o/p_fold -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -d, -f1-4,6,7 | cut -c1-260
find $o -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $o -name "*.db" -delete
