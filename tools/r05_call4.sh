#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r05c4_km.txt
for G in 0 450 225 120 60; do
  timeout 300 python tools/probe_kmeans.py 0 km_group_mb=$G 2>&1 | tail -1 >> $OUT/r05c4_km.txt
done
for R in 64 16; do
  timeout 300 python tools/probe_kmeans.py 0 km_group_mb=225 km_seg_rows=$R 2>&1 | tail -1 >> $OUT/r05c4_km.txt
done
cat $OUT/r05c4_km.txt
