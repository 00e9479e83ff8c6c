#!/bin/bash
# the random call streams of tests/test_gpu_virtual.py over more seeds and longer streams (one-off shake-out, not part of the suite)
#   gpu_fuzz_virtual.sh [first seed] [last seed] [steps per stream]
export TMPDIR=/tmp
o=gpurun_out/fuzz; mkdir -p $o
for seed in $(seq ${1:-1} ${2:-8}); do
  PHYHIP_FUZZ_SEED=$seed PHYHIP_FUZZ_ITERS=${3:-1200} timeout 600 python -m pytest tests/test_gpu_virtual.py -q -k random_call_streams > $o/seed$seed.log 2>&1
  echo "seed $seed: $(grep -E 'passed|failed' $o/seed$seed.log | tail -1)"; grep -E "^FAILED" $o/seed$seed.log | head -5
done
