#!/bin/bash
# cfg2 under 1 / 2 / 4 lanes per pattern (diag build): step, kernel
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag
for g in 2 4 1 2 4; do
  PHYHIP_NT_GROUPS=$g timeout 600 python bench.py --steps 40 > $O/bench_g$g.json 2> $O/bench_g$g.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r04/bench_g$g.json"))
print("G=$g", round(d["value"]), round(d["ms_per_step"]*1e3, 2), round(d["roofline"]["kernel_avg_us"], 2), d["extra"]["cfg4_nt_100x1M_one_gpu"]["ms_per_step"])
PY
done
