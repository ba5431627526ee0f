R=$PWD; export TMPDIR=/tmp; export MILLIEYE_TUNE_CACHE=/tmp/tune_sq.json
python bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 3 --warmup 1 > /dev/null 2>&1
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace -d /tmp/sqq -o s -- python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 3 --warmup 1 --prewarm-seconds 0.3 > /tmp/sqq.log 2>&1
python $R/tools/prof_summary.py /tmp/sqq/s_results.db --pmc | grep "conv_igemm_buf_f32<64, 64, 2, 2, 1, 0, 0, [01], 0>" | awk '{print $(NF-5), $(NF-3), $(NF-2), $(NF-1), $NF}'
