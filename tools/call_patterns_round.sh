#!/bin/bash
# tools/call_patterns_round.sh : the round's call-pattern evidence in ONE gpurun call -- per-phase statistics of the resident commands
# (gpurun_out/r06/call_patterns.txt -> profiles/r06_call_patterns_raw.txt), then tools/profile_round.sh r06 and tools/profile_r05_extra.sh r06
export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print("  us/candidate %.2f  calls %d  dlk %d served %s"%(d["us_per_candidate"], d["surface_calls"], d["dlk"], d["served_by_resident_workgroups"]))'
mkdir -p gpurun_out/r06
{
for hp in "" "--host-pmat"; do
echo "== cfg5 SPR candidates $hp (tools/bench_spr.py --candidates 3000 $hp, PHYHIP_RESIDENT_STATS=1)"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_spr.py --candidates 3000 $hp 2> gpurun_out/r06/stats.err | grep "^{" | python -c "$P"; grep -E "resident|big resident, mean" gpurun_out/r06/stats.err
done
echo "== cfg5 dLk alone (BENCH_DLK_ONLY=1 tools/bench_dlk.py)"; BENCH_DLK_ONLY=1 PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_dlk.py 2>&1 | grep -E "resident|big resident, mean|^\{" | cut -c1-230
echo "== 37x429 aa"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 3000 2> gpurun_out/r06/stats.err | grep "^{" | python -c "$P"; grep -E "resident|20-state res|inside the eval" gpurun_out/r06/stats.err
echo "== 37x429 aa + brlen"; timeout 300 python tools/bench_spr.py --taxa 37 --patterns 429 --states 20 --candidates 2000 --opt-every 4 | grep "^{" | python -c "$P"
echo "== 54x382 nt"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 3000 2> gpurun_out/r06/stats.err | grep "^{" | python -c "$P"; grep -E "^resident" gpurun_out/r06/stats.err
echo "== 54x382 nt host PMat()"; PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 3000 --host-pmat 2> gpurun_out/r06/stats.err | grep "^{" | python -c "$P"; grep -E "^resident" gpurun_out/r06/stats.err
echo "== l2 reuse probe"; timeout 60 phyml_amd/lib/l2_reuse_probe
} > gpurun_out/r06/call_patterns.txt 2>&1
cat gpurun_out/r06/call_patterns.txt | tail -60
bash tools/profile_round.sh r06 2>&1 | tail -30
bash tools/profile_r05_extra.sh r06 2>&1 | tail -15
