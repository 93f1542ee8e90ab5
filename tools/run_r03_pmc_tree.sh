#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
CMD="python $ROOT/tests/devtools/probe_tree.py time floaters 10000000 1"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_tree1 -o pmc -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_tree2 -o pmc -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_tree3 -o pmc -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_tree4 -o pmc -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_tree5 -o pmc -- $CMD > /dev/null 2>&1
cd $ROOT
python tools/rocpd_summary.py --pmc $OUT/pmc_tree1/pmc_results.db $OUT/pmc_tree2/pmc_results.db $OUT/pmc_tree3/pmc_results.db $OUT/pmc_tree4/pmc_results.db $OUT/pmc_tree5/pmc_results.db > $OUT/pmc_r03_tree.txt 2>&1
grep -E "knn_leaf|knn_tree" $OUT/pmc_r03_tree.txt | cut -c1-150
