#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
timeout 600 python tools/probe_k.py 16 25 50 --clouds blobs,floaters --steps 5 --param tree_cand_limit=1024,2048,4096,8192,32768
} > $OUT/r05c25.txt 2>&1
cat $OUT/r05c25.txt
