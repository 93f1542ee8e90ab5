#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
GSX_LIB_PATH=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants/libgsx_hip_treeprof.so timeout 300 python tools/probe_k.py 16 --clouds blobs --steps 1 > $OUT/r05c45.txt 2>&1
grep "slow descent" $OUT/r05c45.txt | sort -u | sort -t' ' -k3 -n -r | head -8 | cut -c1-330
grep -c "slow descent" $OUT/r05c45.txt
