#!/bin/bash
# Update_Eigen_Lr as a command of the resident short-launch evaluator: gate tests, then per-call latency of the recorded
# searches against the previous build (phyml_amd/lib_base), twice each on one box.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_switches.py tests/test_gpu_cases.py tests/test_gpu_search.py -x -q -m gpu > gpurun_out/tests_eig.log 2>&1
grep -E "passed|failed|Error" gpurun_out/tests_eig.log | tail -5
for rep in 1 2; do
for lib in lib_base lib; do
  echo "== $lib"
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1
  PHYHIP_LIBDIR=$R/phyml_amd/$lib timeout 300 python tools/bench_trace.py trace_nucleic_spr 2>/dev/null | tail -1
done
done
PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>&1 | tail -8
