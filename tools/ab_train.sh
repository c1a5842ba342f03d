#!/bin/bash
# Same-box A/B of the ForwardTransformer training bench: BatchNorm sums folded into producer / consumer launches against
# the two-launch form (ESTTS_BENCH_FOLD_SUMS=0), alternating (clocks drift over a minute of load by a few percent).
# Run on the GPU box from the repository root; writes gpurun_out/ab_*.log.
set -e
mkdir -p gpurun_out
run() {   # $1 = tag, $2 = ESTTS_BENCH_FOLD_SUMS
    ESTTS_BENCH_FOLD_SUMS=$2 timeout -k 10 200 python bench.py --workload train --batch 32 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/ab_$1_train.log 2>&1
    echo "$1 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$1_train.log)"
}
for i in 1 2 3 4 5; do run two$i 0; run fold$i 1; done
