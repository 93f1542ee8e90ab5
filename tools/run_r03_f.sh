#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/run_variants_prebuilt.sh "bs8hb2w6 bs4hb2w6 bs8hb2w5 bs8hb1w6 bs4hb4w5" "--steps 20 --no-cpu-baseline --no-secondary" r03f_10m
timeout 300 python bench.py --steps 20 --warmup 5 --exchange slab --no-cpu-baseline > gpurun_out/r03f_slab.json 2> gpurun_out/r03f_slab.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03f_slab.json").read().strip().splitlines()[-1])
print("slab one rank", d["ms_per_step"], d["kernel_ms_per_step"])
PY
for lanes in 8 16; do
python - $lanes <<'PY'
import sys, importlib
sys.path.insert(0, ".")
import bench
gsx = importlib.import_module("3dgsconverter_amd"); L = gsx._lib
ctx = L.Context(0)
r = bench.run_kmeans(L, ctx, gsx, 10_000_000, 3, 1, cpu=False, lanes=int(sys.argv[1]))
print("kmeans lanes", sys.argv[1], r["ms_per_step"])
PY
done 2>&1 | grep "kmeans lanes"
