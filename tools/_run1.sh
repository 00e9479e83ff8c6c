export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f  calls %d  dlk %d served %s"%(d["us_per_candidate"], d["surface_calls"], d["dlk"], d["served_by_resident_workgroups"]))'
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_cfg5.py tests/test_gpu_replay.py tests/test_gpu_trace.py -x -q 2>&1 | tail -6
for hp in "" "--host-pmat"; do
echo "54x382 $hp:"; timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 3000 $hp | grep "^{" | python -c "$P"
echo "54x382 brlen $hp:"; timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 3000 --opt-every 4 $hp | grep "^{" | python -c "$P"
echo "500x100k $hp:"; timeout 300 python tools/bench_spr.py --candidates 2000 $hp | grep "^{" | python -c "$P"
done
echo "host-pmat, residents off:"; PHYHIP_RESIDENT=0 timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 3000 --host-pmat | grep "^{" | python -c "$P"
PHYHIP_RESIDENT=0 timeout 300 python tools/bench_spr.py --candidates 2000 --host-pmat | grep "^{" | python -c "$P"
