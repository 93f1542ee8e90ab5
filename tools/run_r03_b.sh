#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/run_variants_prebuilt.sh "bs8w4 bs8w4hb2 bs4w4 bs4w5 bs8w5" "--steps 20 --no-cpu-baseline --no-secondary" r03b_10m
tools/run_variants_prebuilt.sh "bs8w4 bs4w4 bs4w5" "--steps 50 --n 1000000 --extent 10 --no-cpu-baseline --no-secondary" r03b_1m
tools/run_variants_prebuilt.sh "bs8k33w3 bs4k33w3 bs4k33w2" "--steps 5 --warmup 2 --n 50000000 --extent 10 --k 32 --no-cpu-baseline --no-secondary" r03b_50m
tools/run_variants_prebuilt.sh "bs8w4k9w4 bs4w4k9w4" "--steps 20 --k 8 --no-cpu-baseline --no-secondary" r03b_k8
