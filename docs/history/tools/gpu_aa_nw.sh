repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
for nw in 15 9 7 5; do for v in r3 r2; do
PHYHIP_AA_NW=$nw PHYHIP_LIBDIR=$repo/phyml_amd/lib_$v timeout 200 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --no-cpu-baseline --no-extra --no-call-latency 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v nw $nw', d['config']['patterns_per_gpu'], 'kernel_us', round(d['roofline']['kernel_avg_us'],1), 'frac', round(d['roofline']['frac'],3))"
done; done
for nw in 10 5; do for v in r3 r2; do
PHYHIP_AA_NW=$nw PHYHIP_LIBDIR=$repo/phyml_amd/lib_$v timeout 200 python bench.py --workload cfg3_aa_200x10k --no-cpu-baseline --no-extra --no-call-latency 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v nw $nw', d['config']['patterns_per_gpu'], 'kernel_us', round(d['roofline']['kernel_avg_us'],1), 'frac', round(d['roofline']['frac'],3))"
done; done
