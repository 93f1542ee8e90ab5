#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
timeout 300 python tools/probe_k.py 57 64 --clouds blobs,floaters --steps 5 --param tree_leaf_cap=128,160,192,256
timeout 900 python -m pytest tests/test_sor_tree_gpu.py tests/test_sor_fuzz_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
} > $OUT/r05c47.txt 2>&1
cut -c1-130 $OUT/r05c47.txt
