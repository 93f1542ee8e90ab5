#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_dens -o trace -- python $ROOT/tools/probe_density.py > $OUT/r05c17_dens.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT/prof_dens/trace_results.db > $OUT/r05c17_kernel_stats_density.txt 2>&1
rm -rf $OUT/prof_dens
grep -E "status|host clock" $OUT/r05c17_dens.log; head -16 $OUT/r05c17_kernel_stats_density.txt
