#!/bin/bash
# One gpurun call: smoke, bench (1 GPU), rocprofv3 kernel-trace stats + PMC passes of the same command.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r01}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_${R}.json 2> gpurun_out/bench_${R}.err; echo "bench exit $?" >> gpurun_out/bench_${R}.err
timeout 600 python bench.py --n 10000000 --extent 5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${R}_10m.json 2>> gpurun_out/bench_${R}.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${R} -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${R}_stdout.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_${R} -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_${R}_stdout.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write_${R} -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_write_${R}_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/smoke.log | tail -3
cat gpurun_out/bench_${R}.json gpurun_out/bench_${R}_10m.json
tail -5 gpurun_out/bench_${R}.err
find gpurun_out/prof_${R} -name "*stats*" | head; 
f=$(find gpurun_out/prof_${R} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
du -sh gpurun_out
