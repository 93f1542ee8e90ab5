#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/probe_tree.py check 2>&1 | tail -2
for cfg in "floaters 10000000" "clustered 10000000" "clustered 1000000" "uniform 10000000"; do set -- $cfg; timeout 300 python tools/probe_tree.py time $1 $2 1 2>&1 | grep -E "tree:|step|rror" | cut -c1-300; done
