#!/bin/bash
# round 5, GPU call 1: parity of the new build, headline bench, binning / stats / tree variants (prebuilt here).
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/r05c1_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r05c1_bench.json 2> $OUT/r05c1_bench.err
tools/run_variants_prebuilt.sh "bp6k bp4k_t1024 bp8k_t1024 bp12k_t1024 cswave0" "--steps 20 --no-cpu-baseline --no-secondary" r05c1_grid > /dev/null 2>&1
# 50M k=32 (configs[3] on one GPU): bucket size matters differently there
for tag in default bp8k_t1024 bp12k_t1024; do
  lib=3dgsconverter_amd/variants/libgsx_hip_$tag.so; [ "$tag" = default ] && lib=3dgsconverter_amd/libgsx_hip.so
  GSX_LIB_PATH=$PWD/$lib timeout 300 python bench.py --n 50000000 --extent 10 --k 32 --steps 5 --no-cpu-baseline --no-secondary 2>>$OUT/r05c1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag 50M k32 ms/step %.3f' % d['ms_per_step'], d['kernel_ms_per_step'])" >> $OUT/r05c1_grid_summary.txt 2>&1
done
# tree path: floaters / six blobs at 10M, k = 16 and 25
: > $OUT/r05c1_tree.txt
for tag in default leafw4 leafhb2 leafw4_25w3; do
  lib=3dgsconverter_amd/variants/libgsx_hip_$tag.so; [ "$tag" = default ] && lib=3dgsconverter_amd/libgsx_hip.so
  for C in "floaters 16" "clustered 16" "clustered 25" "clustered 32"; do
    set -- $C
    echo "== $tag $1 k=$2" >> $OUT/r05c1_tree.txt
    GSX_LIB_PATH=$PWD/$lib PROBE_K=$2 timeout 200 python tests/devtools/probe_tree.py time $1 10000000 1 2>&1 | grep -E "step|knn_leaf|near|query" | tail -4 >> $OUT/r05c1_tree.txt
  done
done
cat $OUT/r05c1_pytest.log | tail -5; cut -c1-600 $OUT/r05c1_bench.json; echo; cat $OUT/r05c1_grid_summary.txt; cat $OUT/r05c1_tree.txt
