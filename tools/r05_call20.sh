#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
timeout 300 python tools/probe_k.py 16 25 36 50 --clouds blobs --steps 10
} > $OUT/r05c20.txt 2>&1
cat $OUT/r05c20.txt
