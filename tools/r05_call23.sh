#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_sor_tree_gpu.py tests/test_sor_gpu.py tests/test_sor_fuzz_gpu.py tests/test_dist_gpu.py -m gpu -x -q > $OUT/r05c23_full.txt 2>&1
grep -E "passed|failed|Error|error" $OUT/r05c23_full.txt | tail -5 > $OUT/r05c23.txt
cat $OUT/r05c23.txt
