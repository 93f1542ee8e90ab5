#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line with its secondary configurations (summary printed)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; tail -5 gpurun_out/r03_gpu_tests.log | cut -c1-250
timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -3 gpurun_out/r03_bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["kernel_ms_per_step"])
for k,v in d.get("configs",{}).items():
    print(k, {kk: v.get(kk) for kk in ("ms_per_step","value","error","knn_kernel_ms")})
PY
