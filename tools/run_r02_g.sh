#!/bin/bash
# round 2, call G: slab pipeline through a one-rank RCCL communicator at 10M, K-Means bench line, full GPU suite
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > $OUT/r02g_pytest.log; tail -4 $OUT/r02g_pytest.log
timeout 600 python bench.py --exchange slab --steps 20 > $OUT/r02g_bench_slab1.json 2> $OUT/r02g.err; tail -3 $OUT/r02g.err
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $OUT/r02g_bench_direct.json 2>> $OUT/r02g.err
timeout 900 python bench.py --workload kmeans > $OUT/r02g_bench_kmeans.json 2>> $OUT/r02g.err; tail -3 $OUT/r02g.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02g_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), d["config"].get("parallelism"), d.get("cpu_baseline"), d["roofline"]["frac"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f).read()[-300:])
PY
