#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
GSX_LIB_PATH=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants/libgsx_hip_treeprof.so timeout 300 python tools/probe_k.py 16 --clouds floaters,blobs --steps 1
} > $OUT/r05c26.txt 2>&1
grep -c "slow descent" $OUT/r05c26.txt; grep "slow descent" $OUT/r05c26.txt | sort -t' ' -k3 -n -r | head -30; grep -v "slow descent" $OUT/r05c26.txt | tail -5
