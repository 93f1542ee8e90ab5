#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
echo "== WCAP 24"; timeout 300 python tools/probe_k.py 33 36 41 45 50 57 64 --clouds uniform --steps 10
echo "== WCAP 32 for lists > 32"; GSX_LIB_PATH=$V/libgsx_hip_wcap32.so timeout 300 python tools/probe_k.py 33 36 41 45 50 57 64 --clouds uniform --steps 10
} > $OUT/r05c29.txt 2>&1
cat $OUT/r05c29.txt
