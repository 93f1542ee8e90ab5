#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -25 ) > $OUT/r05c2_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05c2_bench.json 2> $OUT/r05c2_bench.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 8 > $OUT/r05c2_bench_lanes8.json 2>> $OUT/r05c2_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_r05c2_km -o trace -- python $GRAFT_REPO_ROOT/tools/probe_kmeans.py > $GRAFT_REPO_ROOT/$OUT/r05c2_prof_km.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_r05c2_km/trace_results.db > $OUT/r05c2_kernel_stats_km.txt 2>&1
rm -rf $OUT/prof_r05c2_km
tail -8 $OUT/r05c2_pytest.log
python - <<'PY'
import json
for f in ("gpurun_out/r05c2_bench.json","gpurun_out/r05c2_bench_lanes8.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["configs"]["config4"]
        print(f, "headline", d["ms_per_step"], "config4", c.get("ms_per_step"), c.get("batched"), c.get("kernel_ms_per_step"), c.get("roofline",{}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -20 $OUT/r05c2_kernel_stats_km.txt; tail -5 $OUT/r05c2_bench.err
