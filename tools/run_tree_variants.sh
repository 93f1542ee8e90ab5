#!/bin/bash
# GPU box: time prebuilt variants of the tree path (tools/build_variants.sh with ONLY=sor_tree.hip) with tests/devtools/probe_tree.py
#   tools/run_tree_variants.sh "<tags>" "<kind> <n>;<kind> <n>;..."
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAGS=${1:-}
IFS=';' read -ra CFGS <<< "${2:-uniform 10000000}"
timeout 600 python tests/devtools/probe_tree.py check 2>&1 | tail -2
for tag in default $TAGS; do
  lib=3dgsconverter_amd/variants/libgsx_hip_$tag.so
  [ "$tag" = default ] && lib=3dgsconverter_amd/libgsx_hip.so
  [ -f $lib ] || { echo "$tag MISSING"; continue; }
  for cfg in "${CFGS[@]}"; do
    echo "$tag: $(GSX_LIB_PATH=$PWD/$lib timeout 300 python tests/devtools/probe_tree.py time $cfg 1 2>&1 | grep -E "step|rror" | cut -c1-330)"
  done
done
