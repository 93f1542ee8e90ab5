#!/bin/bash
# K-Means assign A/B: centroid-stationary (default) vs streaming matrix-core kernel; parity tests first, kernel trace last
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for V in "" "kmeans_cs=0"; do
  timeout 300 python bench.py --workload kmeans --no-cpu-baseline ${V:+--param $V} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$V]', d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"
done
bash tools/run_trace.sh kmcs "--workload kmeans --no-cpu-baseline --steps 1 --warmup 1" 2>/dev/null | head -8 | cut -c1-120
rm -rf gpurun_out/prof_kmcs
