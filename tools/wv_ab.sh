#!/bin/bash
# A/B on one GPU box (devices differ by several per cent), alternating, three rounds: two option sets of tools/bench_wavernn.py,
# or with WL_BASE=path/to/libestts.so that build against the in-tree one.
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"
for i in 1 2 3; do
  WL_LIB=$WL_BASE timeout -k 10 200 python -u tools/bench_wavernn.py $A 2>&1 | grep "us per step" | sed 's/^/A: /' || exit 1
  timeout -k 10 200 python -u tools/bench_wavernn.py $B 2>&1 | grep "us per step" | sed 's/^/B: /' || exit 1
done
