#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
echo "== 24"; timeout 300 python tools/probe_k.py 18 20 23 25 27 30 32 --clouds uniform --steps 10
echo "== 28"; GSX_LIB_PATH=$V/libgsx_hip_wmid28.so timeout 300 python tools/probe_k.py 18 20 23 25 27 30 32 --clouds uniform --steps 10
echo "== 32"; GSX_LIB_PATH=$V/libgsx_hip_wmid32.so timeout 300 python tools/probe_k.py 18 20 23 25 27 30 32 --clouds uniform --steps 10
} > $OUT/r05c33.txt 2>&1
cut -c1-200 $OUT/r05c33.txt | sed 's/, bin.*fallback queries/ fq/; s/, survivors.*//'
