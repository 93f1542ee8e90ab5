#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/3dgsconverter_amd/variants
{
timeout 900 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2 3; do
echo -n "merge prio: "; PROBE_LEVEL=2 timeout 300 python tools/probe_kmeans.py 0 2>&1 | tail -1 | cut -c1-120
echo -n "no prio: "; GSX_LIB_PATH=$V/libgsx_hip_noprio.so PROBE_LEVEL=2 timeout 300 python tools/probe_kmeans.py 0 2>&1 | tail -1 | cut -c1-120
done
for lv in 5 8; do
echo -n "level $lv merge prio: "; PROBE_LEVEL=$lv timeout 300 python tools/probe_kmeans.py 0 2>&1 | tail -1 | cut -c1-120
echo -n "level $lv no prio: "; GSX_LIB_PATH=$V/libgsx_hip_noprio.so PROBE_LEVEL=$lv timeout 300 python tools/probe_kmeans.py 0 2>&1 | tail -1 | cut -c1-120
done
timeout 600 python tests/devtools/fuzz_kmeans.py 2>&1 | tail -1
} > $OUT/r05c43.txt 2>&1
cat $OUT/r05c43.txt
