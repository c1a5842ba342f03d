#!/bin/bash
# A/B of two builds of the library on one GPU box: tools/_exp/libestts_base.so against the in-tree one, alternating, three rounds.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  WL_LIB=tools/_exp/libestts_base.so timeout -k 10 200 python -u tools/bench_wavernn.py 2>&1 | grep "us per step" | sed 's/^/base: /' || exit 1
  timeout -k 10 200 python -u tools/bench_wavernn.py 2>&1 | grep "us per step" | sed 's/^/new:  /' || exit 1
done
