#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py -m gpu -x -q 2>&1 | tail -3 ) > $OUT/r05c12_km.txt
timeout 300 python tools/probe_kmeans.py 2>&1 | tail -1 | cut -c1-160 >> $OUT/r05c12_km.txt
timeout 300 python tools/probe_kmeans.py 2>&1 | tail -1 | cut -c1-160 >> $OUT/r05c12_km.txt
cat $OUT/r05c12_km.txt
