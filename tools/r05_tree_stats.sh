#!/bin/bash
# the tree path's per-kernel profile alone (the loop of tools/run_round_artifacts.sh), after a change that touches only sor_tree.hip
set -u
export TMPDIR=/tmp
R=${1:-r05}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
for C in "clustered 1000000 16" "floaters 10000000 16" "clustered 10000000 16" "clustered 10000000 25" "clustered 10000000 50"; do
  set -- $C
  PROBE_K=$3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_tree_$1_$2_k$3 -o trace -- python $ROOT/tests/devtools/probe_tree.py time $1 $2 1 > $OUT/tree_trace_${R}_$1_$2_k$3.log 2>&1
  python $ROOT/tools/rocpd_summary.py $OUT/prof_${R}_tree_$1_$2_k$3/trace_results.db > $OUT/kernel_stats_${R}_tree_$1_$2_k$3.txt 2>&1
  rm -rf $OUT/prof_${R}_tree_$1_$2_k$3
done
