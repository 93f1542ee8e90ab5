#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rf in 0.9 1.0 1.1 1.25; do for cfg in "floaters 10000000" "clustered 10000000"; do set -- $cfg; echo "RF=$rf $(GSX_TREE_RF=$rf timeout 300 python tests/devtools/probe_tree.py time $1 $2 1 2>&1 | grep -E "tree:|step" | cut -c1-260 | tr '\n' ' ')"; done; done
