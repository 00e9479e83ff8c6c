export TMPDIR=/tmp
echo "dlk only:"; BENCH_DLK_ONLY=1 PHYHIP_RESIDENT_STATS=1 timeout 300 python tools/bench_dlk.py 2>&1 | grep -E "mean of|^\{|from command" | cut -c1-220
echo "dlk:"; timeout 300 python tools/bench_dlk.py 2>&1 | grep -E "^\{" | cut -c1-220
