#!/bin/bash
# One gpurun call (round 4): parity of the changed kernels, direct + slab one-rank benches, knn_brick PMC traffic.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-b}
timeout 1500 python -m pytest tests/test_sor_gpu.py tests/test_dist_gpu.py -x -q -p no:cacheprovider -k "not 50m" 2>&1 | tail -15 > $O/r04${T}_pytest.log; echo "exit $?" >> $O/r04${T}_pytest.log
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04${T}_bench_direct.json 2> $O/r04${T}_bench_direct.err
timeout 600 python bench.py --exchange slab --steps 30 --warmup 3 --no-cpu-baseline > $O/r04${T}_bench_slab_1rank.json 2> $O/r04${T}_bench_slab_1rank.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_${T}_slab -o trace -- python $GRAFT_REPO_ROOT/bench.py --exchange slab --steps 30 --warmup 4 --no-cpu-baseline > $O/prof_${T}_slab_stdout.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${T}_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_${T}_fetch_stdout.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${T}_write -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_${T}_write_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof_${T}_slab -name "*_results.db" | head -1) > $O/r04${T}_kernel_stats_slab_1rank.txt 2>&1
python tools/rocpd_summary.py --pmc $(find $O/pmc_${T}_fetch $O/pmc_${T}_write -name "*_results.db") > $O/r04${T}_pmc_tcc.txt 2>&1
tail -6 $O/r04${T}_pytest.log
echo ---- direct; python -c "
import json,sys
d=json.load(open('$O/r04${T}_bench_direct.json')); print(d['ms_per_step'], d['kernel_ms_per_step'])
d=json.load(open('$O/r04${T}_bench_slab_1rank.json')); print('slab', d['ms_per_step'], d['kernel_ms_per_step'])"
tail -3 $O/r04${T}_bench_direct.err $O/r04${T}_bench_slab_1rank.err
cat $O/r04${T}_kernel_stats_slab_1rank.txt
grep -i "knn_brick\|bucket_s" $O/r04${T}_pmc_tcc.txt
rm -rf $O/prof_${T}_slab $O/pmc_${T}_fetch $O/pmc_${T}_write
