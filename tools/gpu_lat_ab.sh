#!/bin/bash
# Per-call latency A/B of two builds on the same box: phyml_amd/lib_base against phyml_amd/lib
# (recorded nucleic search prefix, SPR candidates, one evaluation step), parity tests of the new build first.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_cases.py tests/test_gpu_search.py -x -q -m gpu > gpurun_out/tests_lat.log 2>&1
grep -E "passed|failed|Error" gpurun_out/tests_lat.log | tail -3
for rep in 1 2; do
for lib in lib_base lib; do
  echo "== $lib"
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_nucleic_spr 2>/dev/null | tail -1
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 2>/dev/null | tail -1 | cut -c1-300
done
done
for lib in lib_base lib; do
  echo "== $lib"
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg5 us/candidate', round(d['us_per_candidate'],2), 'full ms', round(d['full_both_sides_Lk_ms'],2))"
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  cfg2 step_us', round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"
done
cd /tmp
out=$R/gpurun_out/trace_prof; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/tools/bench_trace.py trace_nucleic_spr device > $out/log.txt 2>&1
for f in $out/*/*kernel_stats.csv; do head -6 $f | cut -c1-200; done
