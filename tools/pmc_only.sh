TAG=r01; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_$TAG.json
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1   # tune
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf_$TAG -o f -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pmf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw_$TAG -o w -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pmw.log 2>&1
python $R/tools/prof_summary.py /tmp/pmf_$TAG/f_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_fetch.txt
python $R/tools/prof_summary.py /tmp/pmw_$TAG/w_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_write.txt
python $R/tools/pmc_traffic.py /tmp/pmf_$TAG/f_results.db /tmp/pmw_$TAG/w_results.db conv_igemm_buf_f32 > $OUT/conv_traffic.json
python $R/tools/pmc_traffic.py /tmp/pmf_$TAG/f_results.db /tmp/pmw_$TAG/w_results.db conv_igemm_buf_h16 > $OUT/conv_traffic_bf16.json
cat $OUT/conv_traffic.json $OUT/conv_traffic_bf16.json
