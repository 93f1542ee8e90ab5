#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sor_gpu.py tests/test_dist_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
tools/run_variants_prebuilt.sh "" "--steps 20 --no-cpu-baseline --no-secondary" r03i_10m
tools/run_variants_prebuilt.sh "" "--steps 50 --n 1000000 --extent 10 --no-cpu-baseline --no-secondary" r03i_1m
