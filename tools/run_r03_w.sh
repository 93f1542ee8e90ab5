#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_sor_tree_gpu.py -x -q -m gpu > gpurun_out/tree_tests.log 2>&1
tail -12 gpurun_out/tree_tests.log | cut -c1-250
timeout 1500 python -m pytest tests/test_sor_gpu.py -x -q -m gpu -k "floaters or inflated or random_cloud or clustered or degenerate or drop_in" 2>&1 | tail -8 | cut -c1-250
