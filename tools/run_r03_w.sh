#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 31 32 33; do timeout 1500 python tests/devtools/fuzz_more.py 100 $seed > gpurun_out/fuzzmore_$seed.log 2>&1; tail -1 gpurun_out/fuzzmore_$seed.log; grep -E "MISMATCH|Error|error|Traceback" gpurun_out/fuzzmore_$seed.log | head -6 | cut -c1-250; done
