export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f  calls %d  dlk %d served %s"%(d["us_per_candidate"], d["surface_calls"], d["dlk"], d["served_by_resident_workgroups"]))'
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_shard_threads.py tests/test_gpu_mixture.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -k "sharded" 2>&1 | tail -3
for th in 0 1; do
echo "54x764 two shards on device 0, threads=$th:"; PHYHIP_SHARD_THREADS=$th timeout 300 python tools/bench_spr.py --taxa 54 --patterns 764 --candidates 3000 --opt-every 4 --devices 0,0 | grep "^{" | python -c "$P"
echo "54x764 two shards, threads=$th, host pmat:"; PHYHIP_SHARD_THREADS=$th timeout 300 python tools/bench_spr.py --taxa 54 --patterns 764 --candidates 3000 --opt-every 4 --devices 0,0 --host-pmat | grep "^{" | python -c "$P"
done
