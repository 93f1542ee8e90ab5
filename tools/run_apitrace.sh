#!/bin/bash
# kernel trace of the one-rank slab pipeline: which small copies / fills does a step enqueue?
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_api -o trace -- python $GRAFT_REPO_ROOT/bench.py --exchange slab --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/prof_api.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_api/trace_results.db | head -30
tail -1 $OUT/prof_api.log | cut -c1-200
