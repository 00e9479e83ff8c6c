#!/bin/bash
# the random call streams of tests/test_gpu_virtual.py over more seeds and longer streams (one-off shake-out, not part of the suite)
export TMPDIR=/tmp
o=gpurun_out/fuzz; mkdir -p $o
for seed in 1 2 3 4 5 6 7 8; do
  PHYHIP_FUZZ_SEED=$seed PHYHIP_FUZZ_ITERS=1200 timeout 600 python -m pytest tests/test_gpu_virtual.py -q -k random_call_streams > $o/seed$seed.log 2>&1
  echo "seed $seed: $(grep -E 'passed|failed' $o/seed$seed.log | tail -1)"; grep -E "^FAILED" $o/seed$seed.log | head -5
done
