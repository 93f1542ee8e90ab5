#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
echo "== cap28 (default build)"; timeout 300 python tools/probe_k.py 10 18 36 41 50
echo "== nocap4"; GSX_LIB_PATH=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants/libgsx_hip_nocap4.so timeout 300 python tools/probe_k.py 10 18 36 41 50
echo "== cap28 again"; timeout 300 python tools/probe_k.py 10 18 36
echo "== parity"; timeout 900 python -m pytest tests/test_sor_gpu.py tests/test_sor_tree_gpu.py -m gpu -x -q 2>&1 | tail -3
} > $OUT/r05c19.txt 2>&1
cat $OUT/r05c19.txt
