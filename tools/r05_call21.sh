#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
echo "== parity"; timeout 900 python -m pytest tests/test_sor_tree_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "== sweep"
timeout 600 python tools/probe_k.py 16 25 36 50 --clouds blobs,floaters --steps 6 --param tree_leaf_cap=64,96,128,192,256
} > $OUT/r05c21.txt 2>&1
cat $OUT/r05c21.txt
