#!/bin/bash
# 20-state short launches (an SPR candidate, Lk(b)): gate tests, then per-call latency of the recorded proteic search and of
# synthetic SPR candidates against the previous build (phyml_amd/lib_base, tools/build_variant.sh base) on one box.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_switches.py tests/test_gpu_cases.py tests/test_gpu_search.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/tests_aalat.log 2>&1
grep -E "passed|failed|Error" gpurun_out/tests_aalat.log | tail -5
for rep in 1 2; do
for lib in lib_base lib; do
  echo "== $lib"
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_proteic_spr device 2>/dev/null | tail -1
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_proteic_spr 2>/dev/null | tail -1
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py --taxa 40 --patterns 1500 --states 20 2>/dev/null | tail -1 | cut -c1-200
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_spr.py --taxa 60 --patterns 6000 --states 20 2>/dev/null | tail -1 | cut -c1-200
done
done
