#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tests/devtools/probe_tree.py check 2>&1 | tail -2
for cfg in "floaters 10000000" "clustered 10000000" "clustered 1000000" "floaters 1000000"; do set -- $cfg; timeout 300 python tests/devtools/probe_tree.py time $1 $2 1 2>&1 | grep -E "step|rror" | cut -c1-300; done
timeout 900 python -m pytest tests/test_sor_tree_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-200
