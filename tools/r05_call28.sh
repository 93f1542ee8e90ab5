#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
{
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d['configs'].items()})"; done
timeout 300 python tools/probe_kmeans.py 2>&1 | tail -6
} > $OUT/r05c28.txt 2>&1
cat $OUT/r05c28.txt
