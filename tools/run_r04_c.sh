#!/bin/bash
# One gpurun call: the whole GPU suite + the default bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-c}
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 2>&1 | tail -60 > $O/r04${T}_pytest_gpu.log; echo "exit $?" >> $O/r04${T}_pytest_gpu.log
#timeout 900 python bench.py > $O/r04${T}_bench.json 2> $O/r04${T}_bench.err; echo "exit $?" >> $O/r04${T}_bench.err
tail -45 $O/r04${T}_pytest_gpu.log
cat > /dev/null <<PY
import json
d=json.load(open("$O/r04${T}_bench.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
for k,v in d["configs"].items():
    print(k, v.get("error") or (v.get("ms_per_step") or v.get("ms_per_call"), v.get("value"), v.get("kernel_ms_per_step")))
PY
tail -3 $O/r04${T}_bench.err
