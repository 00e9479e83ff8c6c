#!/bin/bash
# round 5: what a lone wave per SIMD costs against two (is cfg2's 1.5 waves per SIMD a quantisation loss?), and the four-lanes-per-pattern wave
export TMPDIR=/tmp
o=gpurun_out/r5j; mkdir -p $o
run() { # label, groups, patterns
  PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag PHYHIP_NT_GROUPS=$2 timeout 200 python bench.py --workload cfg2_nt_100x50k --patterns $3 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-companion > $o/b_$1.json 2> $o/b_$1.err
  python -c "
import json; d=json.load(open('$o/b_$1.json')); r=d['roofline']; print('$1 groups $2 patterns $3: kernel %.1f us, step %.1f us, waves %d' % (r['kernel_avg_us'], d['ms_per_step']*1e3, ($3+63)//64*$2))"
}
run g2_1 2 32768
run g2_2 2 65536
run g2_15 2 49152
run g4_1 4 16384
run g4_2 4 32768
run g4_3 4 49152
run g1_1 1 65536
run g2_3 2 98304
