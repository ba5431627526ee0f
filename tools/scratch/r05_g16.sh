cd /root/repo; mkdir -p gpurun_out/r05p; O=gpurun_out/r05p; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" >> $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
(timeout 300 python tools/wgrad_bench.py 8 bf16; timeout 300 python tools/wgrad_bench.py 8) 2>&1 | grep -v amdgpu > $O/wgrad_bench_bf16_vs_f32.txt
export MILLIEYE_TUNE_CACHE=/tmp/tune_r05.json
R=$PWD; DATE=$(date +%Y-%m-%d)
python bench.py --steps 20 --warmup 5 > $O/r05_bench_full_b32.json 2> $O/r05_bench_full_b32.err
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 3 --warmup 1"
C32="$CMD --no-bf16-line --prewarm-seconds 0.3"; C16="$C32 --dtype bf16"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- $CMD > /tmp/kt.log 2>&1
python $R/tools/prof_summary.py /tmp/kt/k_results.db > $R/$O/r05_bench_full_b32_kernel_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf -o f -- $C32 > /tmp/pmf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw -o w -- $C32 > /tmp/pmw.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf16 -o f -- $C16 > /tmp/pmf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw16 -o w -- $C16 > /tmp/pmw16.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmf/f_results.db /tmp/pmw/w_results.db conv_igemm_buf_f32 cfg=yolov3 size=416 batch=32 workload=full dtype=f32 date=$DATE > $R/$O/conv_traffic.json
python $R/tools/pmc_traffic.py /tmp/pmf16/f_results.db /tmp/pmw16/w_results.db conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws cfg=yolov3 size=416 batch=32 workload=full dtype=bf16 date=$DATE > $R/$O/conv_traffic_bf16.json
python $R/tools/pmc_layers.py /tmp/pmf/f_results.db /tmp/pmw/w_results.db 32 416 f32 > $R/$O/r05_layer_traffic_f32.txt 2>&1
python $R/tools/pmc_layers.py /tmp/pmf16/f_results.db /tmp/pmw16/w_results.db 32 416 bf16 > $R/$O/r05_layer_traffic_bf16.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kt16 -o k -- python $R/bench.py --workload detector_train --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 > /tmp/kt16.log 2>&1
python $R/tools/prof_summary.py /tmp/kt16/k_results.db > $R/$O/r05_bf16_bench_detector_train_b8_kernel_stats.txt 2>&1
cd $R
tail -3 $O/gpu_tests.txt; tail -1 $O/smoke.txt; cat $O/conv_traffic.json; cut -c1-250 $O/r05_bench_full_b32.json; cat $O/wgrad_bench_bf16_vs_f32.txt | head -40
