#!/bin/bash
# tools/gpu_aa_ab.sh <libdir names...> : 20-state kernel time of several builds (phyml_amd/lib_<name>; "lib" = the product) on one box
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
for rep in 1 2; do
for v in "$@"; do
  d=$repo/phyml_amd/lib_$v; [ "$v" = lib ] && d=$repo/phyml_amd/lib
  for p in 10000 100000; do
    PHYHIP_LIBDIR=$d timeout 200 python bench.py --workload cfg3_aa_200x10k --patterns $p --no-cpu-baseline --no-extra --no-call-latency 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['patterns_per_gpu'], 'step_ms', round(d['ms_per_step'],4), 'kernel_us', round(d['roofline']['kernel_avg_us'],1), 'frac', round(d['roofline']['frac'],3), d.get('lnL_rel_err'))"
  done
done
done
