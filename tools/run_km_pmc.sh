#!/bin/bash
# SQ counters of the K-Means kernels (one SOG chunk iteration = 156250 x 45, K = 1024)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
P="--workload kmeans --steps 1 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA -d $OUT/pmc_km1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS -d $OUT/pmc_km2 -o pmc -- python $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_km3 -o pmc -- python $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --pmc $OUT/pmc_km1/pmc_results.db $OUT/pmc_km2/pmc_results.db $OUT/pmc_km3/pmc_results.db > $OUT/pmc_r02_kmeans_sq.txt 2>&1
grep -E "kmeans_assign_mfma" $OUT/pmc_r02_kmeans_sq.txt | cut -c1-160
