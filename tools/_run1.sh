export TMPDIR=/tmp
B="python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000"
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f"%d["us_per_candidate"])'
for wg in 1; do
echo "resident, stats of workgroup $wg:"; PHYHIP_RESIDENT_STATS=$wg timeout 120 $B 2> gpurun_out/aa_res_stats_$wg.err | python -c "$P"
grep -E "inside|20-state res|from command" gpurun_out/aa_res_stats_$wg.err
done
for rep in 1 2; do
echo "lib:"; timeout 120 $B | python -c "$P"
done
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/full_gpu_tests.log
