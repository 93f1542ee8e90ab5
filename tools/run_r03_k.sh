#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in 0.5 0.7 0.85; do
 for cfg in "clustered 1000000" "floaters 10000000" "clustered 10000000" "floaters 1000000"; do
  set -- $cfg
  echo "Q=$q $1 $2: $(GSX_PROBE_Q=$q timeout 300 python tools/probe_adaptive.py $1 $2 3 2>&1 | grep "step" | tail -2 | awk '{print $4}' | tr '\n' ' ')"
 done
done
echo "probe off:"; for cfg in "clustered 1000000" "floaters 10000000" "clustered 10000000" "floaters 1000000"; do set -- $cfg; echo "$1 $2: $(GSX_PROBE_SHRINK=0 timeout 300 python tools/probe_adaptive.py $1 $2 3 2>&1 | grep "step" | tail -2 | awk '{print $4}' | tr '\n' ' ')"; done
