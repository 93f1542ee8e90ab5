#!/bin/bash
# GPU box: time prebuilt variant libraries (tools/build_variants.sh) on the headline bench and verify the winner.
#   tools/run_variants_prebuilt.sh "<tags>" "<bench args>" [tag-for-output]
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
TAGS=${1:-}
ARGS=${2:---steps 20 --no-cpu-baseline --no-secondary}
NAME=${3:-var}
: > $OUT/${NAME}_summary.txt
for tag in default $TAGS; do
  lib=3dgsconverter_amd/variants/libgsx_hip_$tag.so
  [ "$tag" = default ] && lib=3dgsconverter_amd/libgsx_hip.so
  [ -f $lib ] || { echo "$tag MISSING" >> $OUT/${NAME}_summary.txt; continue; }
  GSX_LIB_PATH=$PWD/$lib timeout 300 python bench.py $ARGS > $OUT/${NAME}_tmp.json 2>> $OUT/${NAME}.err
  python - "$tag" $OUT/${NAME}_tmp.json >> $OUT/${NAME}_summary.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k=d["kernel_ms_per_step"]
    c3=d.get("config3") or {}
    print("%-14s ms/step %7.4f knn %7.4f bin %6.4f fb %6.4f st %6.4f surv %d thr %.9g %s" % (sys.argv[1], d["ms_per_step"], k["knn"], k["bin"], k["fallback"], k["stats"], d["survivors_rank0"], d["threshold"],
          ("c3 %.3f knn %.3f" % (c3["ms_per_step"], c3["knn_kernel_ms"])) if "ms_per_step" in c3 else ""))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
cat $OUT/${NAME}_summary.txt
