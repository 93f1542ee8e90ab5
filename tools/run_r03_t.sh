#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tests/devtools/probe_tree.py check 2>&1 | tail -3
for cfg in "clustered 1000000" "floaters 1000000" "uniform 1000000" "clustered 10000000" "floaters 10000000" "uniform 10000000"; do set -- $cfg; timeout 300 python tests/devtools/probe_tree.py time $1 $2 1 2>&1 | grep -E "tree:|step|rror" | cut -c1-400; done
