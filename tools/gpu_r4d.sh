#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
PHYHIP_RESIDENT_DEBUG=1 PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag timeout 300 python -m pytest tests/test_gpu_resident.py -x -q -s -k "large_grid_resident_evaluator_spr" > $O/t_c1.log 2>&1; tail -3 $O/t_c1.log
grep -n "owner" $O/t_c1.log | awk -F'owner' '{print $2}' | sort | uniq -c | head; grep -c "^big dLk" $O/t_c1.log
grep -E "^big" $O/t_c1.log | tail -30 | cut -c1-250
