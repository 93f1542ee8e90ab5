#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_sor_tree_gpu.py tests/test_sor_fuzz_gpu.py tests/test_sor_gpu.py -m gpu -x -q 2>&1 | tail -12 ) > $OUT/r05c7_pytest.log
timeout 300 python tests/devtools/probe_tree.py check > $OUT/r05c7_check.txt 2>&1
: > $OUT/r05c7_tree.txt
for G in 1 0; do
  for C in "floaters 16" "clustered 16" "clustered 25" "clustered 32" "clustered 50"; do
    set -- $C
    echo "== group=$G $1 k=$2" >> $OUT/r05c7_tree.txt
    PROBE_K=$2 timeout 200 python tests/devtools/probe_tree.py time $1 10000000 1 tree_near_group=$G 2>&1 | grep -E "step|near:|fallback queries" | tail -3 >> $OUT/r05c7_tree.txt
  done
done
tail -4 $OUT/r05c7_pytest.log; tail -3 $OUT/r05c7_check.txt; cat $OUT/r05c7_tree.txt
