export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f  calls %d  dlk %d"%(d["us_per_candidate"], d["surface_calls"], d["dlk"]))'
echo "cfg5 resident:"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_spr.py --candidates 3000 2> gpurun_out/big_stats.err | python -c "$P"
grep -vE "^\s*$" gpurun_out/big_stats.err | head -40
echo "dlk:"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_dlk.py 2>&1 | tail -30
echo "stamps cfg2:"; PHYHIP_LIBDIR=phyml_amd/lib_diag PHYHIP_ABLATE=8 timeout 300 python bench.py --workload cfg2_nt_100x50k --steps 10 --warmup 3 2>&1 | grep -E "^step" | head -70
