#!/bin/bash
# timing-only ablations of the 20-state kernel (diag build; results invalid): PHYHIP_ABLATE = 256 + bits
repo=${GRAFT_REPO_ROOT:-/root/repo}; cd $repo
export PHYHIP_LIBDIR=$repo/phyml_amd/lib_diag
for a in 0 257 258 260 264 265 272 274 261 269 283 287; do
  for p in 10000 100000; do
    PHYHIP_ABLATE=$a timeout 200 python bench.py --workload cfg3_aa_200x10k --patterns $p --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-call-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', $a - 256 if $a else 0, d['config']['patterns_per_gpu'], 'kernel_us', round(d['roofline']['kernel_avg_us'],1))"
  done
done
