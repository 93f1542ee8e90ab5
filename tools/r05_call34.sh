#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
for lv in 2 5 8; do PROBE_LEVEL=$lv timeout 300 python tools/probe_kmeans.py 2>&1 | tail -1; done
cd /tmp
for lv in 5 8; do
PROBE_LEVEL=$lv timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_km$lv -o trace -- python $GRAFT_REPO_ROOT/tools/probe_kmeans.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$OUT/prof_km$lv/trace_results.db | head -16
rm -rf $GRAFT_REPO_ROOT/$OUT/prof_km$lv
done
} > $OUT/r05c34.txt 2>&1
cat $OUT/r05c34.txt
