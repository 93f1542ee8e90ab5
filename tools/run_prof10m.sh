#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_sor_gpu.py -m gpu -x -q -p no:cacheprovider -k "stats or golden_small or 1m" 2>&1 | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof10m -o trace -- python $GRAFT_REPO_ROOT/bench.py --n 10000000 --extent 5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof10m.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT/prof10m.log
python tools/rocpd_summary.py gpurun_out/prof10m/trace_results.db | head -22
