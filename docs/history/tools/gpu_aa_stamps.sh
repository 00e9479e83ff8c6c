#!/bin/bash
# s_memtime stamps of the first 64 operations of one consumer wave of the 20-state kernel (diag build, PHYHIP_ABLATE=8):
# cfg3 and 100 000 patterns.  Segments per step: 0 entry (issue of the next operation's loads + record requests) | 1 operand
# selection | 2 all-ones test | 3 wait for the ring | 4 matrix phase | 5 product / maximum | 6 rescale + stores.
export TMPDIR=/tmp
out=gpurun_out/aa_stamps; mkdir -p $out
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag
PHYHIP_ABLATE=8 timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 8 --warmup 2 --no-cpu-baseline --no-extra > $out/s8.json 2>$out/stamps_cfg3.txt
PHYHIP_ABLATE=8 timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 8 --warmup 2 --no-cpu-baseline --no-extra > $out/s8b.json 2>$out/stamps_100k.txt
grep "^step" $out/stamps_cfg3.txt | head -70
