#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python tools/probe_k.py 20 25 28 --clouds blobs,floaters --steps 6 --param tree_leaf_cap=64,72,80,88 > $OUT/r05c48.txt 2>&1
cut -c1-125 $OUT/r05c48.txt
