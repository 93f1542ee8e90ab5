#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
timeout 600 python tools/probe_k.py 41 43 44 45 47 48 49 50 52 55 57 58 59 60 62 63 64 --clouds uniform --steps 6
timeout 900 python -m pytest tests/test_sor_gpu.py tests/test_sor_fuzz_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
} > $OUT/r05c32.txt 2>&1
cut -c1-200 $OUT/r05c32.txt | sed 's/, bin.*fallback queries/ fq/; s/, survivors.*//'
