#!/bin/bash
# randomised parity sweeps of the final build (tests/devtools/*: the oracle is the checker), fresh seeds
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
( timeout 900 python tests/devtools/fuzz_parity.py 160 51 2>&1 | tail -4 ) > $OUT/r05_fuzz.txt
( timeout 900 python tests/devtools/fuzz_more.py 60 52 2>&1 | tail -4 ) >> $OUT/r05_fuzz.txt
( timeout 900 python tests/devtools/fuzz_large.py 14 53 2>&1 | tail -4 ) >> $OUT/r05_fuzz.txt
( timeout 900 python tests/devtools/fuzz_kmeans.py 100 54 2>&1 | tail -4 ) >> $OUT/r05_fuzz.txt
cat $OUT/r05_fuzz.txt
