#!/bin/bash
# GPU check after a kernel change: the whole -m gpu suite, then the three bench configurations
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 > $OUT/n_tests.log
timeout 300 python bench.py --steps 30 --no-cpu-baseline > $OUT/n_10m.json 2>$OUT/n.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --n 50000000 --k 32 --extent 8.55 --no-secondary > $OUT/n_50m.json 2>>$OUT/n.err
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-secondary --exchange slab > $OUT/n_slab.json 2>>$OUT/n.err
cat $OUT/n_tests.log
python - <<'PY'
import json,os
for f in ("n_10m","n_50m","n_slab"):
    d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d.get("kernel_ms_per_step"), (d.get("secondary") or {}).get("ms_per_step"))
PY
