#!/bin/bash
# SQ / TCC counters of one 16-bit conv layer (GPU box): bash tools/pmc_h16.sh r01
TAG=${1:-rXX}; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d /tmp/p16a -o a -- python $R/tools/conv16_one.py 14 20 > /tmp/p16a.log 2>&1
python $R/tools/prof_summary.py /tmp/p16a/a_results.db --pmc | grep "conv_igemm\|^#\|^kernel" > $OUT/${TAG}_h16_conv3x3_128to256_b32_pmc_sq.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/p16b -o b -- python $R/tools/conv16_one.py 14 20 > /tmp/p16b.log 2>&1
python $R/tools/prof_summary.py /tmp/p16b/b_results.db --pmc | grep "conv_igemm\|^#\|^kernel" > $OUT/${TAG}_h16_conv3x3_128to256_b32_pmc_tcc.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p16c -o c -- python $R/tools/conv16_one.py 14 20 > /tmp/p16c.log 2>&1
python $R/tools/prof_summary.py /tmp/p16c/c_results.db --pmc | grep "conv_igemm\|^#\|^kernel" > $OUT/${TAG}_h16_conv3x3_128to256_b32_pmc_inst.txt
tail -3 /tmp/p16c.log
cat $OUT/${TAG}_h16_conv3x3_128to256_b32_pmc_*.txt
