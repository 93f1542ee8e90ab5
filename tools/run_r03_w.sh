#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python tests/devtools/fuzz_large.py 30 61 > gpurun_out/fuzzlarge_61.log 2>&1; tail -32 gpurun_out/fuzzlarge_61.log | cut -c1-200
