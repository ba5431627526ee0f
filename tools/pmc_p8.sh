#!/bin/bash
# SQ / LDS / TCC counters of one 16-bit 3x3 layer on one tile id (GPU box): bash tools/pmc_p8.sh TAG TILE H CIN COUT
# three separate --pmc passes (kernel-trace only), summaries -> gpurun_out/prof/${TAG}_p8_*.txt
TAG=${1:-rXX}; TILE=${2:-221}; H=${3:-52}; CIN=${4:-128}; COUT=${5:-256}
R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
N=t${TILE}_${H}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/pa_$N -o a -- python $R/tools/conv16_one.py $TILE 20 $H $CIN $COUT > /tmp/pa.log 2>&1
python $R/tools/prof_summary.py /tmp/pa_$N/a_results.db --pmc | grep "conv3x3\|conv_igemm\|^kernel" > $OUT/${TAG}_p8_${N}_sq.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/pb_$N -o b -- python $R/tools/conv16_one.py $TILE 20 $H $CIN $COUT > /tmp/pb.log 2>&1
python $R/tools/prof_summary.py /tmp/pb_$N/b_results.db --pmc | grep "conv3x3\|conv_igemm" >> $OUT/${TAG}_p8_${N}_sq.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pc_$N -o c -- python $R/tools/conv16_one.py $TILE 20 $H $CIN $COUT > /tmp/pc.log 2>&1
python $R/tools/prof_summary.py /tmp/pc_$N/c_results.db --pmc | grep "conv3x3\|conv_igemm" >> $OUT/${TAG}_p8_${N}_sq.txt
tail -2 /tmp/pc.log
cat $OUT/${TAG}_p8_${N}_sq.txt | cut -c60-200
