#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for spg in 50 200; do
 for i in 1 2; do timeout -k 10 120 python tools/bench_wavernn.py --steps-per-graph $spg 2>&1 | tail -1 | sed "s/.*audio/spg=$spg fused/"; done
done
timeout -k 10 120 python tools/bench_wavernn.py --unfused 2>&1 | tail -1 | sed "s/.*audio/unfused/"
