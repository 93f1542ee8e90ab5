#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for V in "-DGSX_KM_PF=1 -DGSX_KM_PT=1" "-DGSX_KM_PF=2 -DGSX_KM_PT=1 -DGSX_KM_WAVES=2" "-DGSX_KM_PF=1 -DGSX_KM_PT=1 -DGSX_KM_WAVES=1" "-DGSX_KM_PF=2 -DGSX_KM_PT=1"; do
  GSX_EXTRA_FLAGS="$V" python 3dgsconverter_amd/build.py > /dev/null 2>&1 || { echo "$V BUILD FAILED"; continue; }
  bash tools/run_trace.sh kmv "--workload kmeans --no-cpu-baseline --steps 2 --warmup 1" > /dev/null 2>&1
  echo "[$V] $(grep assign_mfma gpurun_out/kernel_stats_kmv.txt | awk '{print $(NF-1)}') us"
done
