#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
echo "== parity"; timeout 1500 python -m pytest tests/test_sor_tree_gpu.py tests/test_sor_gpu.py tests/test_sor_fuzz_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "== k sweep, auto capacity"
timeout 600 python tools/probe_k.py 16 25 27 32 36 41 50 64 --clouds blobs,floaters --steps 6
} > $OUT/r05c22.txt 2>&1
cat $OUT/r05c22.txt
