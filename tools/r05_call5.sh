#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/r05c5_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r05c5_bench.json 2> $OUT/r05c5_bench.err
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --n 2000000 --n3 4000000 > $OUT/r05c5_bench_gpus2.json 2>> $OUT/r05c5_bench.err
tail -5 $OUT/r05c5_pytest.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05c5_bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], d["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "wall", d.get("bench_wall_s"))
for k,v in d["configs"].items():
    print(k, v.get("ms_per_step"), v.get("value"), v.get("error"), (v.get("roofline") or {}).get("frac"), str(v.get("cpu_baseline",{}).get("value")))
try:
    g=json.loads(open("gpurun_out/r05c5_bench_gpus2.json").read().strip().splitlines()[-1])
    print("gpus2", g["ms_per_step"], g.get("phases_ms_per_step"), g.get("cpu_baseline",{}).get("value"), g["config3"].get("speedup_vs_one_gpu"))
except Exception as e:
    print("gpus2 ERR", e)
PY
tail -5 $OUT/r05c5_bench.err
