#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_km1 -o pmc -- python $ROOT/tools/probe_kmeans.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/pmc_km2 -o pmc -- python $ROOT/tools/probe_kmeans.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d $OUT/pmc_km3 -o pmc -- python $ROOT/tools/probe_kmeans.py > /dev/null 2>&1
cd $ROOT
python tools/rocpd_summary.py --pmc $OUT/pmc_km1/pmc_results.db $OUT/pmc_km2/pmc_results.db $OUT/pmc_km3/pmc_results.db 2>&1 | grep -E "^#|^kernel|assign_mfma_cs|segment_sum|exact_list" > $OUT/r05c14_pmc_km.txt
rm -rf $OUT/pmc_km1 $OUT/pmc_km2 $OUT/pmc_km3
cat $OUT/r05c14_pmc_km.txt
