#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag
for gen in 2 1; do for abl in 0 4; do
  PHYHIP_AA_GEN=$gen PHYHIP_ABLATE=$abl timeout 300 python bench.py --workload cfg3_aa_200x10k --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r02d/b_${gen}_${abl}.json 2>gpurun_out/r02d/err.txt
  python -c "import json;d=json.load(open('gpurun_out/r02d/b_${gen}_${abl}.json'));print('gen$gen ablate$abl cfg3',d['ms_per_step'],d['roofline']['kernel_avg_us'])"
  PHYHIP_AA_GEN=$gen PHYHIP_ABLATE=$abl timeout 300 python bench.py --workload cfg3_aa_200x10k --patterns 100000 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r02d/c_${gen}_${abl}.json 2>>gpurun_out/r02d/err.txt
  python -c "import json;d=json.load(open('gpurun_out/r02d/c_${gen}_${abl}.json'));print('gen$gen ablate$abl 100k',d['ms_per_step'],d['roofline']['kernel_avg_us'])"
done; done
