#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py -m gpu -x -q 2>&1 | tail -25 ) > $OUT/r05c3_pytest.log
timeout 300 python tools/probe_kmeans.py > $OUT/r05c3_km.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_r05c3_km -o trace -- python $GRAFT_REPO_ROOT/tools/probe_kmeans.py > $GRAFT_REPO_ROOT/$OUT/r05c3_prof_km.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$OUT/calib_fetch -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > $GRAFT_REPO_ROOT/$OUT/r05c3_calib.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$OUT/calib_write -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_r05c3_km/trace_results.db > $OUT/r05c3_kernel_stats_km.txt 2>&1
python tools/rocpd_summary.py --pmc $OUT/calib_fetch/pmc_results.db $OUT/calib_write/pmc_results.db > $OUT/r05c3_calib_pmc.txt 2>&1
rm -rf $OUT/prof_r05c3_km $OUT/calib_fetch $OUT/calib_write
tail -6 $OUT/r05c3_pytest.log; cat $OUT/r05c3_km.txt | tail -2; head -14 $OUT/r05c3_kernel_stats_km.txt; cat $OUT/r05c3_calib.txt | tail -9; cat $OUT/r05c3_calib_pmc.txt
