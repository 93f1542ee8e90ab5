#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
for cfg in "floaters 10000000"; do
  set -- $cfg
  rm -rf $OUT/prof_tree_$1
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_tree_$1 -o p -- python $GRAFT_REPO_ROOT/tests/devtools/probe_tree.py time $1 $2 1 > $OUT/prof_tree_$1.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/prof_tree_$1/p_results.db 2>&1 | head -24 | cut -c1-150
done
