#!/bin/bash
# One iteration of WaveRNN kernel work on the GPU box: the WaveRNN parity tests, then per-kernel times of the config-4 pass.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_wavernn.py -x -q --timeout 120 --timeout-method thread > gpurun_out/wv.log 2>&1 || { tail -30 gpurun_out/wv.log; exit 1; }
tail -2 gpurun_out/wv.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_wv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wv -- python3 $R/tools/bench_wavernn.py "$@" > $R/gpurun_out/prof_wv.log 2>&1
grep "us per step" $R/gpurun_out/prof_wv.log
f=$(ls -t $(find $R/gpurun_out/prof_wv -name "*kernel_stats.csv") | head -1)
head -9 $f | python3 -c "import csv,sys; [print(r[0][-60:], r[1], r[3][:8]) for r in csv.reader(sys.stdin)]"
