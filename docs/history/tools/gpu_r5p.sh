#!/bin/bash
# round 5, stress: the diag build with PHYHIP_VIRT_MIN_OPS=3 (phyhip_set_virtual_buffers(3) from the environment: every list of three or
# more operations leaves its tip x tip results virtual -- also the path updates of a tree search, which the default threshold of 16
# never touches) under the parity tests and under PhyML's real searches in check mode.  Tests that assert COUNTS of the default
# threshold may fail here; values must not.
export TMPDIR=/tmp
o=gpurun_out/r5p; mkdir -p $o
export PHYHIP_LIBDIR=$PWD/phyml_amd/lib_diag LD_LIBRARY_PATH=$PWD/phyml_amd/lib_diag:$LD_LIBRARY_PATH PHYHIP_VIRT_MIN_OPS=3
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_switches.py --deselect tests/test_gpu_bench_cmd.py > $o/full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -30
timeout 600 python tools/search_check.py 80 4000 2>&1 | tail -1 | cut -c1-900 | tee $o/check_nt_80x4000.json
timeout 600 python tools/search_check.py 40 1500 --aa 2>&1 | tail -1 | cut -c1-900 | tee $o/check_aa_40x1500.json
timeout 600 python tools/search_check.py 80 4000 --device-pmat 2>&1 | tail -1 | cut -c1-900 | tee $o/check_nt_80x4000_device_pmat.json
python - <<'PY'
import ctypes, os
print("library in use:", os.environ["PHYHIP_LIBDIR"])
PY
