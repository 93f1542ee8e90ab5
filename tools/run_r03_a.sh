#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r03a_pytest.log
tail -5 gpurun_out/r03a_pytest.log
tools/run_variants_prebuilt.sh "base sqrt bs8 bs8w4 bs8gs gs bs8hb8 rf5 rf4 bs8rf5" "--steps 20 --no-cpu-baseline --no-secondary" r03a_10m
tools/run_variants_prebuilt.sh "bs8 bs8k33w3 bs8w4k33w3" "--steps 6 --no-cpu-baseline --no-secondary --config3" r03a_c3
for m in 6.25 6.5 7.0; do
  tools/run_variants_prebuilt.sh "bs8 bs8w4" "--steps 20 --no-cpu-baseline --no-secondary --param grid_points_per_cell=$m" r03a_m$m
done
