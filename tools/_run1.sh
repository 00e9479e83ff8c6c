export TMPDIR=/tmp
for dp in 0 1; do
echo "search 54x382 nt, GLUE_DEVICE_PMAT=$dp:"; GLUE_DEVICE_PMAT=$dp timeout 600 python tools/search_bench.py 54 382 --skip-host 2>&1 | tail -1 | cut -c1-330
echo "search 40x300 aa, GLUE_DEVICE_PMAT=$dp:"; GLUE_DEVICE_PMAT=$dp timeout 900 python tools/search_bench.py 40 300 --skip-host --aa 2>&1 | tail -1 | cut -c1-330
done
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_resident.py tests/test_gpu_trace.py tests/test_gpu_replay.py tests/test_gpu_switches.py -x -q 2>&1 | tail -6
