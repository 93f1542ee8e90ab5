#!/bin/bash
# rocprofv3 kernel-trace stats of the bench at 10M splats
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_10m -o trace -- python $GRAFT_REPO_ROOT/bench.py --n 10000000 --extent 5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_10m.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_10m/trace_results.db | head -20
