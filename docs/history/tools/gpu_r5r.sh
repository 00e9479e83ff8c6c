#!/bin/bash
# round 5: the register-blocked 20-state matrix kernel (pmat20_kernel): same bits as pmat_kernel<20> (switch tests on the diag
# build), the 20-state parity tests, and its duration -- three matrices of an SPR candidate, 397 of a whole tree -- against the old one
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/r5r; mkdir -p $o
cd $R
timeout 600 python -m pytest tests/test_gpu_switches.py -q -k "PMAT or default" > $o/switches.log 2>&1; echo "switches rc=$?"; grep -E "passed|failed" $o/switches.log | tail -1; grep -E "^FAILED|^E  " $o/switches.log | head -10
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixture.py tests/test_gpu_virtual.py tests/test_gpu_replay.py tests/test_gpu_cases.py -q -x > $o/subset.log 2>&1; echo "subset rc=$?"; grep -E "passed|failed" $o/subset.log | tail -1; grep -E "^FAILED|^E  " $o/subset.log | head -10
cd /tmp
stats() { f=$(find $1 -name "*kernel_stats.csv" | head -1); grep -E "pmat" "$f" | cut -d, -f1-4,6,7 | cut -c1-200; }
python $R/tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 2>&1 | tail -1 | cut -c1-330 | tee $o/plain_37x429.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p_new -o aa -- python $R/tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 > $o/p_new.log 2>&1; echo "== product (new kernel), SPR candidates"; stats $o/p_new
export PHYHIP_LIBDIR=$R/phyml_amd/lib_diag
for mode in 0 2; do
  PHYHIP_PMAT20=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $o/p_diag$mode -o aa -- python $R/tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 > $o/p_diag$mode.log 2>&1; echo "== diag PHYHIP_PMAT20=$mode, SPR candidates: $(tail -1 $o/p_diag$mode.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["us_per_candidate"])')"; stats $o/p_diag$mode
  PHYHIP_PMAT20=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $o/t_diag$mode -o aa -- python $R/bench.py --workload cfg3_aa_200x10k --steps 30 --warmup 5 --no-extra --no-companion --no-cpu-baseline > $o/t_diag$mode.log 2>&1; echo "== diag PHYHIP_PMAT20=$mode, cfg3 full Lk (397 matrices per step): $(grep '^{"metric"' $o/t_diag$mode.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_avg_us"])')"; stats $o/t_diag$mode
done
find $o -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $o -name "*.db" -delete
