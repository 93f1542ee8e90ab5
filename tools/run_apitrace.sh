#!/bin/bash
# HIP API + kernel trace of the one-rank slab pipeline (which small copies / fills does a step enqueue?)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats -d $OUT/prof_api -o trace -- python $GRAFT_REPO_ROOT/bench.py --exchange slab --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/prof_api.log 2>&1
ls $OUT/prof_api
python - <<'PY'
import sqlite3,os,glob
db=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_api/*.db")[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'stat' in t.lower() or 'top' in t.lower() or 'summary' in t.lower()][:20])
for t in tabs:
    if t.lower() in ('top','top_kernels','hip_api_stats','memory_copy_stats') or 'api' in t.lower() and 'summ' in t.lower():
        try:
            for r in c.execute(f"select * from {t} limit 25"): print(t, r)
        except Exception as e: print(t, e)
PY
