#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "floaters 10000000" "clustered 10000000"; do set -- $cfg; timeout 300 python tests/devtools/probe_tree.py time $1 $2 1 2>&1 | grep -E "step|rror" | cut -c1-300; done
for cfg in "floaters 10000000" "clustered 10000000"; do set -- $cfg; GSX_LIB_PATH=$PWD/3dgsconverter_amd/variants/libgsx_hip_prof.so timeout 300 python tests/devtools/probe_tree.py time $1 $2 1 2>&1 | grep -E "slow descent" | sort | uniq | sort -k3 -n -r | awk '{print $3, $4, $5, $6, $7, $8, $9,$10,$11,$12,$13,$14,$15,$16,$17,$18,$19,$20,$21,$22,$23,$24}' | uniq -f1 | head -12 | cut -c1-200; done
