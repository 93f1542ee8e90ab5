#!/bin/bash
# GPU box: the randomised parity sweeps (tests/devtools/fuzz_*.py); summaries in gpurun_out/.   usage: tools/run_fuzz.sh [cases] [first seed]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=${1:-150}; S=${2:-100}
for t in parity more kmeans; do
  timeout 2400 python tests/devtools/fuzz_$t.py $C $S > gpurun_out/fuzz_${t}_$S.log 2>&1
  tail -1 gpurun_out/fuzz_${t}_$S.log; grep -E "MISMATCH|Traceback" gpurun_out/fuzz_${t}_$S.log | head -5 | cut -c1-250
done
timeout 2400 python tests/devtools/fuzz_large.py 10 $S > gpurun_out/fuzz_large_$S.log 2>&1; tail -1 gpurun_out/fuzz_large_$S.log
