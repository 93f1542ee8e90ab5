#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_density_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -3 ) > $OUT/r05c18.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_dens -o trace -- python $GRAFT_REPO_ROOT/tools/probe_density.py >> $GRAFT_REPO_ROOT/$OUT/r05c18.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/$OUT/prof_dens/trace_results.db > $OUT/r05c18_kernel_stats_density.txt 2>&1 || python tools/rocpd_summary.py $OUT/prof_dens/trace_results.db > $OUT/r05c18_kernel_stats_density.txt 2>&1
grep -E "passed|failed|status|host clock" $OUT/r05c18.txt; head -12 $OUT/r05c18_kernel_stats_density.txt
