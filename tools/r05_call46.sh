#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
echo "== 28 words"; timeout 300 python tools/probe_k.py 36 41 50 64 --clouds blobs,floaters --steps 6
echo "== 32 words for lists > 32"; GSX_LIB_PATH=$V/libgsx_hip_twbig32.so timeout 300 python tools/probe_k.py 36 41 50 64 --clouds blobs,floaters --steps 6
} > $OUT/r05c46.txt 2>&1
cut -c1-120 $OUT/r05c46.txt
