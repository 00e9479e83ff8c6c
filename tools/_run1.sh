export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f  calls %d  dlk %d served %s"%(d["us_per_candidate"], d["surface_calls"], d["dlk"], d["served_by_resident_workgroups"]))'
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_cfg5.py tests/test_gpu_parity.py tests/test_gpu_cases.py -x -q 2>&1 | tail -4
for rep in 1 2; do
echo "dlk:"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_dlk.py 2>&1 | grep -E "mean of|^\{" | cut -c1-220
echo "500x100k brlen:"; timeout 300 python tools/bench_spr.py --candidates 2000 --opt-every 4 | grep "^{" | python -c "$P"
done
echo "dlk 1M:"; timeout 300 python tools/bench_dlk.py 1000000 2>&1 | grep -E "^\{" | cut -c1-220
echo "dlk 20k:"; timeout 300 python tools/bench_dlk.py 20000 2>&1 | grep -E "^\{" | cut -c1-220
