export TMPDIR=/tmp
bash tools/gpu_full_check.sh 2>&1 | tail -12 | cut -c1-600
for r in 1 0; do
echo "search 54x382 nt, residents=$r:"; PHYHIP_RESIDENT=$r timeout 600 python tools/search_bench.py 54 382 --skip-host 2>&1 | tail -1 | cut -c1-400
echo "search 150x20000 nt, residents=$r:"; PHYHIP_RESIDENT=$r timeout 900 python tools/search_bench.py 150 20000 --skip-host 2>&1 | tail -1 | cut -c1-400
done
