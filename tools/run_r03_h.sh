#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
tools/run_variants_prebuilt.sh "" "--steps 20 --no-cpu-baseline --no-secondary" r03h_10m
tools/run_variants_prebuilt.sh "" "--steps 50 --n 1000000 --extent 10 --no-cpu-baseline --no-secondary" r03h_1m
