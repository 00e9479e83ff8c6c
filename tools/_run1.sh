export TMPDIR=/tmp
for dp in "" "--device-pmat"; do
echo "check 120x6000 nt $dp:"; timeout 900 python tools/search_check.py 120 6000 $dp 2>&1 | tail -1 | cut -c1-420
echo "check 60x1200 aa $dp:"; timeout 900 python tools/search_check.py 60 1200 --aa $dp 2>&1 | tail -1 | cut -c1-420
echo "check 90x500 nt, 3 shards + helper threads $dp:"; GLUE_DEVICES=0,0,0 PHYHIP_SHARD_THREADS=1 timeout 900 python tools/search_check.py 90 500 $dp 2>&1 | tail -1 | cut -c1-420
done
PHYHIP_FUZZ_N=120 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2
