cd /root/repo; mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ws" > gpurun_out/r05a/ws_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05a/ws_tests.txt
for only in "1x1" ; do timeout 600 python tools/conv_bench.py --batch 32 --tiles 3,2,1,50 --splits 1 --only "$only" >> gpurun_out/r05a/ws_bench.txt 2>&1; done
for only in "32->64" "64->128"; do timeout 600 python tools/conv_bench.py --batch 32 --tiles 3,2,1,60 --splits 1 --only "$only" >> gpurun_out/r05a/ws_bench.txt 2>&1; done
for only in "3x3 32->64" "3x3 64->128"; do timeout 600 python tools/conv_bench.py --batch 32 --tiles 3,60 --splits 1 --res --only "$only" >> gpurun_out/r05a/ws_bench_res.txt 2>&1; done
for v in 1 2; do MILLIEYE_WS32_VARIANT=$v timeout 600 python tools/conv_bench.py --batch 32 --tiles 50 --splits 1 --only "1x1 256->128" >> gpurun_out/r05a/ws_bench_var$v.txt 2>&1; MILLIEYE_WS32_VARIANT=$v timeout 600 python tools/conv_bench.py --batch 32 --tiles 60 --splits 1 --only "32->64" >> gpurun_out/r05a/ws_bench_var$v.txt 2>&1; done
MILLIEYE_TUNE_VERBOSE=1 timeout 900 python bench.py --no-cpu-baseline --no-bf16-line --no-batch-sweep --no-accuracy > gpurun_out/r05a/bench_f32.json 2> gpurun_out/r05a/bench_f32.err
MILLIEYE_NO_WS32=50,60 timeout 900 python bench.py --no-cpu-baseline --no-bf16-line --no-batch-sweep --no-accuracy > gpurun_out/r05a/bench_f32_nows.json 2> gpurun_out/r05a/bench_f32_nows.err
tail -3 gpurun_out/r05a/ws_tests.txt; cat gpurun_out/r05a/ws_bench.txt | cut -c1-250
