#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sor_gpu.py -x -q -m gpu -k "rim_queries" 2>&1 | tail -3 | cut -c1-200
for seed in 21 22 23 24 25; do timeout 1200 python tests/devtools/fuzz_parity.py 150 $seed > gpurun_out/fuzz_$seed.log 2>&1; tail -1 gpurun_out/fuzz_$seed.log; grep MISMATCH gpurun_out/fuzz_$seed.log | head -5; done
