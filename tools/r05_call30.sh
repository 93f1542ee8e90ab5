#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
echo "== WCAP 32 for lists > 32, points per cell"; GSX_LIB_PATH=$V/libgsx_hip_wcap32.so timeout 600 python tools/probe_k.py 36 41 45 50 57 64 --clouds uniform --steps 6 --param grid_points_per_cell=13,14,15,16,17,18,19,20,22
} > $OUT/r05c30.txt 2>&1
cat $OUT/r05c30.txt
