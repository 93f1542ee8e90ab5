#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
for c in "5 0.5 uniform" "5 0.9 uniform" "5 1.0 uniform" "5 0.5 floaters" "5 0.9 floaters" "5 1.0 floaters" "5 0.5 blobs" "5 1.0 blobs"; do
  echo "== $c"; timeout 300 python tools/probe_density.py 10000000 $c 2>&1 | tail -2 | cut -c1-200
done
} > $OUT/r05c41.txt 2>&1
cat $OUT/r05c41.txt
