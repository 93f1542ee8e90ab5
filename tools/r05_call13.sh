#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r05c13_km.txt
for tag in default kmml2 kmml4 default kmml2; do
  lib=3dgsconverter_amd/variants/libgsx_hip_$tag.so; [ "$tag" = default ] && lib=3dgsconverter_amd/libgsx_hip.so
  echo "== $tag" >> $OUT/r05c13_km.txt
  GSX_LIB_PATH=$PWD/$lib timeout 300 python tools/probe_kmeans.py 2>&1 | tail -1 | cut -c1-160 >> $OUT/r05c13_km.txt
done
( GSX_LIB_PATH=$PWD/3dgsconverter_amd/variants/libgsx_hip_kmml2.so timeout 600 python -m pytest tests/test_kmeans_gpu.py -m gpu -x -q 2>&1 | tail -2 ) >> $OUT/r05c13_km.txt
cat $OUT/r05c13_km.txt
