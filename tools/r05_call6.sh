#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
for C in "clustered 25" "clustered 16" "floaters 16" "clustered 50"; do
  set -- $C
  PROBE_K=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r05c6_$1_$2 -o trace -- python $ROOT/tests/devtools/probe_tree.py time $1 10000000 1 > $OUT/r05c6_trace_$1_$2.log 2>&1
  python $ROOT/tools/rocpd_summary.py $OUT/prof_r05c6_$1_$2/trace_results.db > $OUT/r05c6_kernel_stats_tree_$1_k$2.txt 2>&1
  rm -rf $OUT/prof_r05c6_$1_$2
  echo "== $1 k=$2"; grep -E "step|near:" $OUT/r05c6_trace_$1_$2.log | tail -2; head -24 $OUT/r05c6_kernel_stats_tree_$1_k$2.txt
done
