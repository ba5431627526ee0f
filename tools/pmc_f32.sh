#!/bin/bash
# SQ counters of one fp32 layer on one tile id (GPU box): bash tools/pmc_f32.sh TAG TILE H CIN COUT K S [res]
# separate --pmc passes (kernel-trace only), summary -> gpurun_out/prof/${TAG}_f32_*.txt
TAG=${1:-rXX}; TILE=${2:-3}; H=${3:-52}; CIN=${4:-256}; COUT=${5:-128}; K=${6:-1}; S=${7:-1}; RES=${8:-}
R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
N=t${TILE}_${H}_${CIN}_${COUT}_k${K}s${S}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/pa_$N -o a -- python $R/tools/conv32_one.py $TILE 20 $H $CIN $COUT $K $S $RES > /tmp/pa.log 2>&1
python $R/tools/prof_summary.py /tmp/pa_$N/a_results.db --pmc | grep "conv\|^kernel" > $OUT/${TAG}_f32_${N}_sq.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace -d /tmp/pc_$N -o c -- python $R/tools/conv32_one.py $TILE 20 $H $CIN $COUT $K $S $RES > /tmp/pc.log 2>&1
python $R/tools/prof_summary.py /tmp/pc_$N/c_results.db --pmc | grep "conv" >> $OUT/${TAG}_f32_${N}_sq.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace -d /tmp/pd_$N -o d -- python $R/tools/conv32_one.py $TILE 20 $H $CIN $COUT $K $S $RES > /tmp/pd.log 2>&1
python $R/tools/prof_summary.py /tmp/pd_$N/d_results.db --pmc | grep "conv" >> $OUT/${TAG}_f32_${N}_sq.txt
tail -2 /tmp/pd.log
cat $OUT/${TAG}_f32_${N}_sq.txt
