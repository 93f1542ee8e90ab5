#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
echo "== default (8 in flight)"; timeout 300 python tools/probe_k.py 16 25 --clouds blobs,floaters --steps 10
echo "== 4 in flight"; GSX_LIB_PATH=$V/libgsx_hip_tqf4.so timeout 300 python tools/probe_k.py 16 25 --clouds blobs,floaters --steps 10
echo "== 16 in flight"; GSX_LIB_PATH=$V/libgsx_hip_tqf16.so timeout 300 python tools/probe_k.py 16 25 --clouds blobs,floaters --steps 10
echo "== profile"; GSX_LIB_PATH=$V/libgsx_hip_treeprof.so timeout 300 python tools/probe_k.py 16 --clouds floaters,blobs --steps 1 2>&1 | sort -t' ' -k3 -n -r | head -12
echo "== parity"; timeout 900 python -m pytest tests/test_sor_tree_gpu.py tests/test_sor_fuzz_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
} > $OUT/r05c27.txt 2>&1
cat $OUT/r05c27.txt
