#!/bin/bash
# One gpurun call (round 4): the torch-free multi-rank path on the one-GPU box -- GPU dist tests (hostwire), bench.py --gpus 2,
# the one-rank slab pipeline (overhead against the direct step) with its kernel trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_dist_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -40 > $O/r04_pytest_dist.log; echo "exit $?" >> $O/r04_pytest_dist.log
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --n 2000000 --n3 4000000 > $O/r04_bench_gpus2.json 2> $O/r04_bench_gpus2.err; echo "exit $?" >> $O/r04_bench_gpus2.err
timeout 600 python bench.py --exchange slab --steps 20 --warmup 3 --no-cpu-baseline > $O/r04_bench_slab_1rank.json 2> $O/r04_bench_slab_1rank.err; echo "exit $?" >> $O/r04_bench_slab_1rank.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04_bench_direct.json 2> $O/r04_bench_direct.err; echo "exit $?" >> $O/r04_bench_direct.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r04_slab -o trace -- python $GRAFT_REPO_ROOT/bench.py --exchange slab --steps 30 --warmup 4 --no-cpu-baseline > $O/prof_r04_slab_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof_r04_slab -name "*_results.db" | head -1) > $O/r04_kernel_stats_slab_1rank.txt 2>&1
tail -15 $O/r04_pytest_dist.log
echo ---- gpus2; cat $O/r04_bench_gpus2.json; tail -5 $O/r04_bench_gpus2.err
echo ---- slab1; cat $O/r04_bench_slab_1rank.json; tail -3 $O/r04_bench_slab_1rank.err
echo ---- direct; cat $O/r04_bench_direct.json; tail -3 $O/r04_bench_direct.err
cat $O/r04_kernel_stats_slab_1rank.txt
rm -rf $O/prof_r04_slab
