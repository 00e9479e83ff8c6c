export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_bench_cmd.py -x -q 2>&1 | tail -6
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06/bench_default.json; cat gpurun_out/r06/bench_default.json | cut -c1-1700
