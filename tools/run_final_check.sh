#!/bin/bash
# GPU box: what the driver runs at the end of a round -- the -m gpu suite, smoke(), the default bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/final_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["kernel_ms_per_step"])
for k,v in d.get("configs",{}).items(): print(k, v.get("ms_per_step"), v.get("value"), v.get("error"))
PY
