#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
echo "== WCAP 32 for lists > 32, points per cell"; GSX_LIB_PATH=$V/libgsx_hip_wcap32.so timeout 600 python tools/probe_k.py 43 45 47 50 52 55 57 60 64 --clouds uniform --steps 6 --param grid_points_per_cell=0,15,16,16.5,20,22,24,26
} > $OUT/r05c31.txt 2>&1
cut -c1-200 $OUT/r05c31.txt | sed 's/uniform grid_points_per_cell=//; s/, bin.*fallback queries/ fq/; s/, survivors.*//'
