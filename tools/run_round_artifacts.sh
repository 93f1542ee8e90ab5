#!/bin/bash
# Produce the round's judged artefacts in ONE gpurun call: bench lines, rocprofv3 kernel-trace stats of the same
# commands, PMC passes (separate passes: TCC FETCH / WRITE, SQ issue counters).  usage: run_round_artifacts.sh r02
set -u
export TMPDIR=/tmp
R=${1:-r06}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py > $OUT/bench_${R}_10m.json 2> $OUT/bench_${R}.err
timeout 600 python bench.py --exchange slab --steps 20 --no-cpu-baseline > $OUT/bench_${R}_slab_1rank.json 2>> $OUT/bench_${R}.err
timeout 600 python bench.py --n 50000000 --extent 10 --k 32 --steps 10 --no-cpu-baseline --no-secondary > $OUT/bench_${R}_50m_k32.json 2>> $OUT/bench_${R}.err
# the N-rank code path on this one-GPU box: bench.py starts its own ranks, which share the GPU over the hostwire transport
# (round 6: the secondary lines run BASELINE's own sizes -- 50M k=32, 10M density -> SOR -- on the GLOBAL seed-0 clouds and check the
#  gathered masks against the reference-run hashes of tests/golden/large_cases.json: mask_matches_reference_run)
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --n 2000000 --no-cpu-baseline > $OUT/bench_${R}_gpus2_hostwire.json 2>> $OUT/bench_${R}.err
cd /tmp
B="--steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R} -o trace -- python $ROOT/bench.py $B > $OUT/prof_${R}.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_km -o trace -- python $ROOT/tools/probe_kmeans.py > $OUT/prof_${R}_km.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_slab -o trace -- python $ROOT/bench.py --exchange slab $B > $OUT/prof_${R}_slab.log 2>&1
# the SOG writer's device-resident core (10M x 248-byte rows -> texel arrays): stage clock + kernel trace of the same command
timeout 600 python $ROOT/tools/probe_sog.py > $OUT/sog_probe_${R}_10m.log 2>&1
PROBE_REPS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_sog -o trace -- python $ROOT/tools/probe_sog.py > $OUT/prof_${R}_sog.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT/prof_${R}_sog/trace_results.db > $OUT/kernel_stats_${R}_sog.txt 2>&1
rm -rf $OUT/prof_${R}_sog
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_dens -o trace -- python $ROOT/tools/probe_density.py > $OUT/prof_${R}_dens.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT/prof_${R}_dens/trace_results.db > $OUT/kernel_stats_${R}_density.txt 2>&1
rm -rf $OUT/prof_${R}_dens
P="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_sq1_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_sq2_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
# the float64 / float32 split of the VALU stream (bench.py prices the issue cycles with it)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -d $OUT/pmc_sq3_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_BF16 -d $OUT/pmc_sq4_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
# what bucket_scatter waits for: write-request stalls at the L2's memory interface, pending-request stalls in the L1
timeout 600 rocprofv3 --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_RDREQ -d $OUT/pmc_ea_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_GATE_EN1 -d $OUT/pmc_tcp_${R} -o pmc -- python $ROOT/bench.py $P > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py --pmc $OUT/pmc_ea_${R}/pmc_results.db $OUT/pmc_tcp_${R}/pmc_results.db $OUT/pmc_grbm_${R}/pmc_results.db 2>&1 | grep -E "^#|^kernel|bucket_|bbox_partial" > $OUT/pmc_${R}_binning.txt
# the clouds a uniform grid is bad at: adaptive mode -> Morton-tree path (csrc/sor_tree.hip)
for C in "clustered 1000000 16" "floaters 10000000 16" "clustered 10000000 16" "clustered 10000000 25" "clustered 10000000 50"; do
  set -- $C
  PROBE_K=$3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_tree_$1_$2_k$3 -o trace -- python $ROOT/tests/devtools/probe_tree.py time $1 $2 1 > $OUT/tree_trace_${R}_$1_$2_k$3.log 2>&1
  python $ROOT/tools/rocpd_summary.py $OUT/prof_${R}_tree_$1_$2_k$3/trace_results.db > $OUT/kernel_stats_${R}_tree_$1_$2_k$3.txt 2>&1
  rm -rf $OUT/prof_${R}_tree_$1_$2_k$3
done
# counters of the tree path's kernels on the 10M scene (separate passes; only the text summary travels back)
TC="python $ROOT/tests/devtools/probe_tree.py time floaters 10000000 1"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_tree1 -o pmc -- $TC > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_tree2 -o pmc -- $TC > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_tree3 -o pmc -- $TC > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_tree4 -o pmc -- $TC > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_tree5 -o pmc -- $TC > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py --pmc $OUT/pmc_tree1/pmc_results.db $OUT/pmc_tree2/pmc_results.db $OUT/pmc_tree3/pmc_results.db $OUT/pmc_tree4/pmc_results.db $OUT/pmc_tree5/pmc_results.db 2>&1 | grep -E "^#|^kernel|knn_leaf|knn_tree|tree_" > $OUT/pmc_${R}_tree.txt
rm -rf $OUT/pmc_tree1 $OUT/pmc_tree2 $OUT/pmc_tree3 $OUT/pmc_tree4 $OUT/pmc_tree5
# counter calibration on known byte counts (tools/ubench/fetch_calib.hip) and the matrix / vector overlap probe
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_fetch_${R} -o pmc -- $ROOT/tools/ubench/fetch_calib > $OUT/calib_${R}.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_write_${R} -o pmc -- $ROOT/tools/ubench/fetch_calib > /dev/null 2>&1
python $ROOT/tools/rocpd_summary.py --pmc $OUT/calib_fetch_${R}/pmc_results.db $OUT/calib_write_${R}/pmc_results.db > $OUT/calib_pmc_${R}.txt 2>&1
rm -rf $OUT/calib_fetch_${R} $OUT/calib_write_${R}
timeout 120 $ROOT/tools/ubench/mfma_valu_overlap > $OUT/mfma_valu_overlap_${R}.txt 2>&1
cd $ROOT
timeout 1200 python bench.py --gpus 8 --steps 5 --warmup 2 --n 1000000 --no-cpu-baseline > $OUT/bench_${R}_gpus8_hostwire.json 2>> $OUT/bench_${R}.err
for T in "" _km _slab; do python tools/rocpd_summary.py $OUT/prof_${R}${T}/trace_results.db > $OUT/kernel_stats_${R}${T}.txt 2>&1; done
python tools/rocpd_summary.py --pmc $OUT/pmc_fetch_${R}/pmc_results.db $OUT/pmc_write_${R}/pmc_results.db > $OUT/pmc_${R}_tcc.txt 2>&1
python tools/rocpd_summary.py --pmc $OUT/pmc_sq1_${R}/pmc_results.db $OUT/pmc_sq2_${R}/pmc_results.db $OUT/pmc_sq3_${R}/pmc_results.db $OUT/pmc_sq4_${R}/pmc_results.db $OUT/pmc_grbm_${R}/pmc_results.db > $OUT/pmc_${R}_sq.txt 2>&1
cut -c1-400 $OUT/bench_${R}_10m.json; echo
head -8 $OUT/kernel_stats_${R}.txt; grep -E "knn_brick" $OUT/pmc_${R}_tcc.txt $OUT/pmc_${R}_sq.txt | cut -c1-170
