R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $R/gpurun_out/sq_counters.txt
run() { # name, counters..., -- cmd
  name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/$name -o $name -- "$@" > /tmp/$name.log 2>&1
  python $R/tools/prof_summary.py /tmp/$name/${name}_results.db --pmc | grep -E "conv_igemm|mix<|counter" >> $R/gpurun_out/pmc_mix.txt
}
rm -f $R/gpurun_out/pmc_mix.txt
CONV="python $R/tools/conv_bench.py --batch 12 --custom 128,128,256,3,1 --tiles 1,5 --reps 20 --prewarm 0.2"
run a1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA -- $CONV
run a2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -- $CONV
run a3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY -- $CONV
run a4 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $CONV
run b1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA -- $R/tools/mfma_mix.bin
run b2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -- $R/tools/mfma_mix.bin
run b3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY -- $R/tools/mfma_mix.bin
run b4 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $R/tools/mfma_mix.bin
tail -3 /tmp/a1.log
