#!/bin/bash
# A/B of compile-time variants of the library on the box: each line of $1 (a file) is a set of -D flags;
# the default build is run first.  usage: run_variants.sh variants.txt "bench args"
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS=${2:---steps 20 --no-cpu-baseline --no-secondary}
TAG=${3:-var}
: > $OUT/${TAG}_summary.txt
run() {
  timeout 300 python bench.py $ARGS > $OUT/${TAG}_tmp.json 2>> $OUT/${TAG}.err
  python - "$1" >> $OUT/${TAG}_summary.txt <<'PY'
import json,sys,os
try:
    d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/%s_tmp.json" % os.environ.get("TAGX","var")).read().strip().splitlines()[-1])
    print("%-60s value %8.1f ms/step %7.4f knn %7.4f bin %6.4f fb %6.4f st %6.4f surv %d" % (sys.argv[1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"]["knn"], d["kernel_ms_per_step"]["bin"], d["kernel_ms_per_step"]["fallback"], d["kernel_ms_per_step"]["stats"], d["survivors_rank0"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
export TAGX=$TAG
run "default"
while IFS= read -r line; do
  [ -z "$line" ] && continue
  GSX_EXTRA_FLAGS="$line" python 3dgsconverter_amd/build.py > /dev/null 2>> $OUT/${TAG}.err || { echo "$line BUILD FAILED" >> $OUT/${TAG}_summary.txt; continue; }
  run "$line"
done < "$1"
cat $OUT/${TAG}_summary.txt
