export TMPDIR=/tmp
bash tools/gpu_full_check.sh 2>&1 | tail -12 | cut -c1-1800
bash tools/call_patterns_round.sh 2>&1 | tail -40 | cut -c1-400
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06/bench_default.json; cat gpurun_out/r06/bench_default.json
