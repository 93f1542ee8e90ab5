#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r05c11_km.txt
for tag in default kmabl1 kmabl2 kmabl3; do
  lib=3dgsconverter_amd/variants/libgsx_hip_$tag.so; [ "$tag" = default ] && lib=3dgsconverter_amd/libgsx_hip.so
  echo "== $tag" >> $OUT/r05c11_km.txt
  GSX_LIB_PATH=$PWD/$lib timeout 300 python tools/probe_kmeans.py 2>&1 | tail -1 | cut -c1-160 >> $OUT/r05c11_km.txt
done
cat $OUT/r05c11_km.txt
