#!/bin/bash
# tools/final_numbers.sh <tag>: the round's headline numbers in one GPU call (same box for all of them) -> gpurun_out/<tag>_numbers/
tag=${1:-r02}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/${tag}_numbers; mkdir -p $out
export TMPDIR=/tmp
cd $repo
python bench.py --steps 100 --warmup 20 > $out/bench_default.json 2> $out/bench_default.err
python bench.py --patterns 125000 --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $out/bench_nt_125k.json 2>/dev/null
python bench.py --patterns 1000000 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench_nt_1M.json 2>/dev/null
python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_aa_100k.json 2>/dev/null
PHYHIP_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_cfg4_one_rank.json
python tools/bench_spr.py 2>/dev/null | tail -1 > $out/bench_spr_cfg5.json
python tools/bench_spr.py --taxa 54 --patterns 382 --candidates 4000 2>/dev/null | tail -1 > $out/bench_spr_54x382.json
python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1 > $out/bench_trace_nucleic.txt
python tools/bench_trace.py trace_proteic_spr device 2>/dev/null | tail -1 > $out/bench_trace_proteic.txt
phyml_amd/lib/membench 2 20 2>/dev/null | tail -1 > $out/membench.json
for f in $out/*.json $out/*.txt; do echo "== $(basename $f)"; head -c 900 $f; echo; done
