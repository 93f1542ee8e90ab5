#!/bin/bash
# points-per-cell sweep of the 10M bench (knn_brick vs ring fallback balance) + 1M + 50M k=32
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/m_summary.txt
row() {
  python - "$1" >> $OUT/m_summary.txt <<'PY'
import json,sys,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/m_tmp.json").read().strip().splitlines()[-1])
print("%-22s ms/step %7.4f knn %7.4f bin %6.4f fb %6.4f st %6.4f surv %d nfb %d" % (sys.argv[1], d["ms_per_step"], d["kernel_ms_per_step"]["knn"], d["kernel_ms_per_step"]["bin"], d["kernel_ms_per_step"]["fallback"], d["kernel_ms_per_step"]["stats"], d["survivors_rank0"], d["grid"]["n_fallback"]))
PY
}
for m in 0 6.25 6.5 6.75 7.0; do
  timeout 200 python bench.py --steps 20 --no-cpu-baseline --no-secondary --param grid_points_per_cell=$m > $OUT/m_tmp.json 2>> $OUT/m.err; row "10M m=$m"
done
for m in 0 6.5 6.75; do
  timeout 200 python bench.py --steps 50 --no-cpu-baseline --n 1000000 --extent 10 --no-secondary --param grid_points_per_cell=$m > $OUT/m_tmp.json 2>> $OUT/m.err; row "1M m=$m"
done
for m in 0 13 13.5 14; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --n 50000000 --k 32 --extent 8.55 --no-secondary --param grid_points_per_cell=$m > $OUT/m_tmp.json 2>> $OUT/m.err; row "50M k=32 m=$m"
done
cat $OUT/m_summary.txt
