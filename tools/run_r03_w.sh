#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 51 52; do timeout 1500 python tests/devtools/fuzz_kmeans.py 150 $seed > gpurun_out/fuzzkm_$seed.log 2>&1; tail -1 gpurun_out/fuzzkm_$seed.log; grep -E "MISMATCH|Error|Traceback" gpurun_out/fuzzkm_$seed.log | head -12 | cut -c1-250; done
