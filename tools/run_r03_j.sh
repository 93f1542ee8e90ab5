#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sor_gpu.py tests/test_density_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -6
for cfg in "clustered 1000000" "floaters 10000000" ; do
  set -- $cfg
  timeout 300 python tools/probe_adaptive.py $1 $2 4 2>&1 | grep -v "^W2\|^E2" | grep "level [0-9]: n\|probed\|step\|surv" | cut -c1-200 | tail -30
done
