#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3 4 7; do
  rm -rf $R/gpurun_out/prof_wv
  ESTTS_WL_DEBUG=$dbg timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wv -- python3 $R/tools/bench_wavernn.py --utts 64 --steps-per-graph 50 > $R/gpurun_out/prof_wv.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_wv -name "*kernel_stats.csv" | head -1)
  echo "debug=$dbg $(grep 'us per step' $R/gpurun_out/prof_wv.log | sed 's/.*audio//')"; grep "wavernn_layer" $f | cut -d, -f1-4 | cut -c40-150
done
