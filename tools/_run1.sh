export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f  calls %d  dlk %d"%(d["us_per_candidate"], d["surface_calls"], d["dlk"]))'
timeout 300 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -5
for rep in 1 2; do
echo "aa resident:"; PHYHIP_RESIDENT_STATS=1 timeout 120 python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 2> gpurun_out/aa_res_stats.err | python -c "$P"
grep -E "from command|inside the eval|20-state res" gpurun_out/aa_res_stats.err
echo "aa resident + brlen:"; timeout 120 python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 2000 --opt-every 4 | python -c "$P"
done
