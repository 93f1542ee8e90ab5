#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/run_variants_prebuilt.sh "bs8w5 bs8w6 bs4w6 pfw3 bs8w4pfw bs8w5pfw bs8w5pfwgs" "--steps 20 --no-cpu-baseline --no-secondary" r03c_10m
tools/run_variants_prebuilt.sh "bs8w5k33w3 bs8w5k33w4" "--steps 5 --warmup 2 --n 50000000 --extent 10 --k 32 --no-cpu-baseline --no-secondary" r03c_50m
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03c_bench_full.json 2> gpurun_out/r03c_bench_full.err ) 2> gpurun_out/r03c_bench_full.time
tail -c 1500 gpurun_out/r03c_bench_full.err
cat gpurun_out/r03c_bench_full.time
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03c_bench_full.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("cpu_baseline"), d.get("bench_wall_s"))
for k,v in d.get("configs",{}).items():
    print(k, {kk:vv for kk,vv in v.items() if kk not in ("roofline","grid","workload")})
PY
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*F64[A-Z0-9_]*" | sort -u > gpurun_out/r03c_counters.txt; wc -l gpurun_out/r03c_counters.txt; head -50 gpurun_out/r03c_counters.txt
