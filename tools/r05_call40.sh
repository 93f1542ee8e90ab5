#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
timeout 900 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for lv in 5 8; do
  for p in kmeans_lds_update=0 kmeans_lds_update=1; do
    echo -n "level $lv $p: "; PROBE_LEVEL=$lv timeout 300 python tools/probe_kmeans.py 0 $p 2>&1 | tail -1 | cut -c1-140
  done
done
timeout 600 python tests/devtools/fuzz_kmeans.py 2>&1 | tail -2
} > $OUT/r05c40.txt 2>&1
cat $OUT/r05c40.txt
