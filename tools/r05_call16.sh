#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r05c16_tree.txt
for F in 0.3 0.4; do
  for C in "clustered 16" "clustered 25" "clustered 50" "floaters 16"; do
    set -- $C
    echo "== cell=$F $1 k=$2" >> $OUT/r05c16_tree.txt
    PROBE_K=$2 timeout 200 python tests/devtools/probe_tree.py time $1 10000000 1 tree_near_cell=$F 2>&1 | grep -E "step|near:" | tail -2 | cut -c1-230 >> $OUT/r05c16_tree.txt
  done
done
cat $OUT/r05c16_tree.txt
