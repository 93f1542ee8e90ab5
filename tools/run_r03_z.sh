#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "floaters 10000000"; do set -- $cfg; GSX_LIB_PATH=$PWD/3dgsconverter_amd/variants/libgsx_hip_prof.so timeout 300 python tests/devtools/probe_tree.py time $1 $2 1 2>&1 | grep -E "slow descent" | sort | uniq | sort -k3 -n -r | awk '{$1="";$2="";print}' | uniq -f2 | head -14 | cut -c1-300; done
