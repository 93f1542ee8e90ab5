#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 11 12 13; do timeout 1200 python tests/devtools/fuzz_parity.py 120 $seed > gpurun_out/fuzz_$seed.log 2>&1; tail -1 gpurun_out/fuzz_$seed.log; grep -c "algo 3" gpurun_out/fuzz_$seed.log; grep MISMATCH gpurun_out/fuzz_$seed.log | head -5; done
