#!/bin/bash
# Same-box A/B of the ForwardTransformer training bench, alternating (clocks drift over a minute of load by a few percent):
# the all-taps weight-gradient kernel against the per-tap one (ESTTS_WGRAD_PER_TAP=1).
# Run on the GPU box from the repository root; writes gpurun_out/ab_*.log.
set -e
mkdir -p gpurun_out
run() {   # $1 = tag, $2 = ESTTS_WGRAD_PER_TAP
    ESTTS_WGRAD_PER_TAP=$2 timeout -k 10 200 python bench.py --workload train --batch 32 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/ab_$1_train.log 2>&1
    echo "$1 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$1_train.log)"
}
for i in 1 2 3; do run pertap$i 1; run alltaps$i 0; done
