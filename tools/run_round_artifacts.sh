#!/bin/bash
# Produce the round's judged artefacts in ONE gpurun call: bench lines (1M default, 10M), rocprofv3
# kernel-trace stats of the same command, PMC FETCH_SIZE / WRITE_SIZE passes (separate passes).
set -u
export TMPDIR=/tmp
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench_${R}.json 2> $OUT/bench_${R}.err
timeout 600 python bench.py --n 10000000 --extent 5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_${R}_10m.json 2>> $OUT/bench_${R}.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R} -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_${R}.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_${R} -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_${R}.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_${R} -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_${R}.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_${R}_10m -o pmc -- python $GRAFT_REPO_ROOT/bench.py --n 10000000 --extent 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_${R}_10m -o pmc -- python $GRAFT_REPO_ROOT/bench.py --n 10000000 --extent 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat $OUT/bench_${R}.json $OUT/bench_${R}_10m.json
python tools/rocpd_summary.py $OUT/prof_${R}/trace_results.db | head -12
python tools/rocpd_summary.py --pmc $OUT/pmc_fetch_${R}/pmc_results.db $OUT/pmc_write_${R}/pmc_results.db $OUT/pmc_fetch_${R}_10m/pmc_results.db $OUT/pmc_write_${R}_10m/pmc_results.db | grep -E "knn_|cell_"
