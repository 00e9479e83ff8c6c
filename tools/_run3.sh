export TMPDIR=/tmp
bash tools/gpu_full_check.sh 2>&1 | tail -12 | cut -c1-1800
timeout 600 python tools/multigpu_selfcheck.py --devices 0,0 --patterns 200000 --steps 10 2>&1 | grep -E "ms_per_step|shard_ms|collective|OK|MISMATCH|call_pattern" 
PHYHIP_SHARD_THREADS=1 timeout 600 python tools/multigpu_selfcheck.py --devices 0,0 --patterns 200000 --steps 10 2>&1 | grep -E "ms_per_step|shard_ms|collective|OK|MISMATCH|call_pattern"
