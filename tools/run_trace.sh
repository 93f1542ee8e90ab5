#!/bin/bash
# rocprofv3 kernel trace of a bench command: run_trace.sh TAG "bench args"
set -u
export TMPDIR=/tmp
TAG=$1; ARGS=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_$TAG/trace_results.db > $OUT/kernel_stats_$TAG.txt 2>&1
head -40 $OUT/kernel_stats_$TAG.txt
