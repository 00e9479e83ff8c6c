#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02e
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag
PHYHIP_ABLATE=8 timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 8 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r02e/s8.json 2>gpurun_out/r02e/stamps_cfg3.txt
PHYHIP_ABLATE=12 timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 8 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r02e/s12.json 2>gpurun_out/r02e/stamps_cfg3_noA.txt
PHYHIP_ABLATE=8 timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 8 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r02e/s8b.json 2>gpurun_out/r02e/stamps_100k.txt
head -40 gpurun_out/r02e/stamps_cfg3.txt
