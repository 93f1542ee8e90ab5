#!/bin/bash
# round 2, call D: full GPU parity suite (incl. the reference-generated K-Means / 10M fixtures) + bench line
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $OUT/r02d_pytest.log
tail -12 $OUT/r02d_pytest.log
timeout 600 python bench.py > $OUT/r02d_bench.json 2> $OUT/r02d_bench.err; tail -2 $OUT/r02d_bench.err
python -c "
import json; d=json.loads(open('$OUT/r02d_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['secondary'], d['cpu_baseline'], d['roofline']['frac'])"
