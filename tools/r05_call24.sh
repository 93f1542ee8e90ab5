#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
GSX_TRACE_LEVELS=1 timeout 600 python tools/probe_k.py 16 25 50 --clouds floaters --steps 1 --param tree_scale=0,1.0,1.09,1.19,1.30,1.41,1.54,1.68,1.83 2>&1 | grep -v "probe:\|near:"
} > $OUT/r05c24.txt 2>&1
cat $OUT/r05c24.txt
