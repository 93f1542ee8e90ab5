#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lanes in 1 2 4 8; do
python - $lanes <<'PY'
import sys, importlib, json, time
sys.path.insert(0, ".")
import bench
gsx = importlib.import_module("3dgsconverter_amd"); L = gsx._lib
ctx = L.Context(0)
r = bench.run_kmeans(L, ctx, gsx, 10_000_000, 3, 1, cpu=False, lanes=int(sys.argv[1]))
print("kmeans lanes", sys.argv[1], r["ms_per_step"], r["kernel_ms_per_step"]["assign (operands + mfma + exact list)"], r["kernel_ms_per_step"]["update (label sort + segmented reduce)"])
PY
done 2>&1 | grep "kmeans lanes" | tee gpurun_out/r03e_kmeans_lanes.txt
cd /tmp
for cfg in "clustered 1000000" "floaters 10000000"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03e_prof_$1 -o p -- python $GRAFT_REPO_ROOT/tools/probe_adaptive.py $1 $2 3 > $GRAFT_REPO_ROOT/gpurun_out/r03e_probe_$1.log 2>&1
  grep -v "^W2\|^E2" $GRAFT_REPO_ROOT/gpurun_out/r03e_probe_$1.log | tail -60
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/r03e_prof_$1 -name "*results.db" | head -1) 2>/dev/null | head -40 | tee $GRAFT_REPO_ROOT/gpurun_out/r03e_stats_$1.txt
done
