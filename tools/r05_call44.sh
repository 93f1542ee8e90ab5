#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
timeout 300 python tools/probe_k.py 16 25 --clouds blobs,floaters --steps 10
timeout 900 python -m pytest tests/test_sor_tree_gpu.py tests/test_sor_fuzz_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
} > $OUT/r05c44.txt 2>&1
cat $OUT/r05c44.txt
