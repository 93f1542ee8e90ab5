#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_sor_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
tools/run_variants_prebuilt.sh "" "--steps 20 --no-cpu-baseline --no-secondary" r03g_10m
tools/run_variants_prebuilt.sh "" "--steps 50 --n 1000000 --extent 10 --no-cpu-baseline --no-secondary" r03g_1m
tools/run_variants_prebuilt.sh "" "--steps 5 --warmup 2 --n 50000000 --extent 10 --k 32 --no-cpu-baseline --no-secondary" r03g_50m
cd /tmp
P="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_g -o pmc -- python $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_g -o pmc -- python $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --pmc $OUT/pmc_fetch_g/pmc_results.db $OUT/pmc_write_g/pmc_results.db 2>&1 | grep "knn_brick_kernel<17, false\|bucket_scatter\|ring_fast" | cut -c1-150
