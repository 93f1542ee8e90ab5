#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-d}
timeout 1200 python -m pytest tests/test_kmeans_gpu.py tests/test_sog_gpu.py tests/test_sor_gpu.py -q -p no:cacheprovider -k "not 50m" 2>&1 | tail -12 > $O/r04${T}_pytest.log
#timeout 900 python bench.py > $O/r04${T}_bench.json 2> $O/r04${T}_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_${T}_km -o trace -- python $GRAFT_REPO_ROOT/tools/probe_kmeans.py > $O/prof_${T}_km.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof_${T}_km -name "*_results.db" | head -1) > $O/r04${T}_kernel_stats_kmeans.txt 2>&1
tail -8 $O/r04${T}_pytest.log
cat > /dev/null <<PY
import json
d=json.load(open("$O/r04${T}_bench.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
for k,v in d["configs"].items():
    print(k, v.get("error") or (v.get("ms_per_step") or v.get("ms_per_call"), v.get("value"), v.get("kernel_ms_per_step")))
PY
cat $O/r04${T}_kernel_stats_kmeans.txt
rm -rf $O/prof_${T}_km
