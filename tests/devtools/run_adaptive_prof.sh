#!/bin/bash
# kernel trace of one refined (adaptive) SOR call on the outlier-inflated scene
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cat > /tmp/adapt_one.py <<PY
import sys, os
sys.path.insert(0, "$GRAFT_REPO_ROOT/tools"); sys.path.insert(0, "$GRAFT_REPO_ROOT")
import gpu_probe as g
from oracle import datasets
ctx = g.L.Context(0)
ctx.set_param("adaptive", 1)
g.run(ctx, datasets.scene_with_floaters(int(sys.argv[1]), 1), 16, 2, 0.0, reps=1, label="scene")
ctx.close()
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_adapt -o trace -- python /tmp/adapt_one.py ${1:-10000000} > $OUT/prof_adapt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_adapt/trace_results.db | head -32
