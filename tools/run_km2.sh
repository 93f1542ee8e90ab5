#!/bin/bash
# K-Means update A/B: segmented sum (default) vs one workgroup per centroid; parity tests first
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for V in "" "-DGSX_KM_SEGSUM=0"; do
  GSX_EXTRA_FLAGS="$V" python 3dgsconverter_amd/build.py > /dev/null 2>&1 || { echo "$V BUILD FAILED"; continue; }
  timeout 300 python bench.py --workload kmeans --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$V]', d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"
done
