#!/bin/bash
# PMC passes over bench.py (separate passes: SQ has 8 slots, TCC 4).  usage: run_pmc.sh TAG "bench args"
set -u
export TMPDIR=/tmp
TAG=${1:-x}; ARGS=${2:---steps 3 --warmup 1 --no-cpu-baseline}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d $OUT/pmc_${TAG}_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_${TAG}_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py --pmc $(ls gpurun_out/pmc_${TAG}_*/pmc_results.db) 2>&1 | grep -E "knn_|counter" | head -80
