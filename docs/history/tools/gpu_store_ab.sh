#!/bin/bash
# result stores with the non-temporal cache policy (the product) against plain stores (lib_st0: tools/build_variant.sh st0 -DPHYHIP_STORE_AUX=0), same box: cfg2, 1 M nt, cfg3, 100 000 aa
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
for rep in 1 2; do for v in lib ${VARIANTS:-lib_st0}; do
  for args in "" "--patterns 1000000 --steps 10 --warmup 3" "--workload cfg3_aa_200x10k" "--workload cfg3_aa_200x10k --patterns 100000 --steps 20 --warmup 5"; do
    PHYHIP_LIBDIR=$repo/phyml_amd/$v timeout 200 python bench.py $args --no-cpu-baseline --no-extra --no-call-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['states'], d['config']['patterns_per_gpu'], 'step_ms', round(d['ms_per_step'],4), 'kernel_us', round(d['roofline']['kernel_avg_us'],1), d.get('lnL_rel_err'))"
  done
done; done
