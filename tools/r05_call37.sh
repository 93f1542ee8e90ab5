#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
for lv in 5 8 2; do
  for t in "" kmabl1 kmabl2 kmabl3; do
    if [ -z "$t" ]; then unset GSX_LIB_PATH; else export GSX_LIB_PATH=$V/libgsx_hip_$t.so; fi
    echo -n "level $lv ${t:-product}: "; PROBE_LEVEL=$lv timeout 300 python tools/probe_kmeans.py 0 2>&1 | tail -1 | cut -c1-110
  done
done
} > $OUT/r05c37.txt 2>&1
cat $OUT/r05c37.txt
