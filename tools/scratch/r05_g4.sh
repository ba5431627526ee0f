cd /root/repo; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" >> $O/gpu_tests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
timeout 600 python tools/bf16_row_diff.py bf16 > $O/bf16_row_diff.txt 2>&1
tail -5 $O/gpu_tests.txt; tail -c 600 $O/bench_default.json; tail -30 $O/bf16_row_diff.txt
