#!/bin/bash
# Runtime knobs that could change launch / dispatch latency; recorded nucleic search prefix, product build.
export TMPDIR=/tmp
run() { echo -n "$1 : "; env $1 timeout 300 python tools/bench_trace.py trace_nucleic_spr device 2>/dev/null | tail -1 | sed 's/.*record, //'; }
for rep in 1 2; do
run "X=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HSA_ENABLE_INTERRUPT=0"
run "GPU_MAX_HW_QUEUES=1"
run "DEBUG_CLR_BLIT_KERNARG_OPT=1"
run "HIP_SKIP_ABORT_ON_GPU_ERROR=1"
run "ROC_SIGNAL_POOL_SIZE=128"
run "HSA_ENABLE_SDMA=0"
run "AMD_SERIALIZE_KERNEL=0"
run "HIP_USE_ADVISE_PREFETCH=0"
run "DEBUG_HIP_KERNARG_COPY_OPT=1"
run "DEBUG_HIP_7_PREVIEW=1"
done
