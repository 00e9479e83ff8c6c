#!/bin/bash
# round 5: larger real searches in check mode with virtual buffers / two wave shapes on (every Lk / dLk of the reference against the device's)
export TMPDIR=/tmp
o=gpurun_out/r5n; mkdir -p $o
timeout 900 python tools/search_check.py 80 4000 2>&1 | tail -1 | cut -c1-900 | tee $o/check_nt_80x4000.json
timeout 900 python tools/search_check.py 60 9000 2>&1 | tail -1 | cut -c1-900 | tee $o/check_nt_60x9000.json
timeout 900 python tools/search_check.py 40 1500 --aa 2>&1 | tail -1 | cut -c1-900 | tee $o/check_aa_40x1500.json
timeout 900 python tools/search_check.py 80 4000 --device-pmat 2>&1 | tail -1 | cut -c1-900 | tee $o/check_nt_80x4000_device_pmat.json
