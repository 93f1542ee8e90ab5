#!/bin/bash
# what the driver runs at round end, in one GPU call: pytest -m gpu, smoke(), bench.py --gpus 1 --steps 20 --warmup 5
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/r05_final_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/r05_final_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_final_bench.json 2> $OUT/r05_final_bench.err
tail -3 $OUT/r05_final_pytest.log; tail -1 $OUT/r05_final_smoke.log | cut -c1-200
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_final_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"].get("binding_resource",{}).get("busy_frac"), "wall", d.get("bench_wall_s"))
print("cpu", d.get("cpu_baseline"))
for k,v in d["configs"].items():
    print(k, v.get("ms_per_step"), v.get("value"), v.get("error"), (v.get("roofline") or {}).get("frac"), str((v.get("cpu_baseline") or {}).get("value")))
PY
