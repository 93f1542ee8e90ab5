#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/r03d_pytest.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/r03d_pytest.log | tail -40
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03d_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
