export TMPDIR=/tmp
bash tools/gpu_full_check.sh 2>&1 | tail -12 | cut -c1-1800
bash tools/_run2.sh 2>&1 | tail -60 | cut -c1-600
