cd /root/repo; mkdir -p gpurun_out/r05u; O=gpurun_out/r05u
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" >> $O/gpu_tests.txt
timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bf16_bench_detector_train_b8.json 2>/dev/null
timeout 900 python bench.py --workload detector_train --batch 8 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_detector_train_b8.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_full_b32.json 2> $O/r05_bench_full_b32.err
tail -3 $O/gpu_tests.txt; cut -c80-170 $O/r05_bf16_bench_detector_train_b8.json $O/r05_bench_detector_train_b8.json; cut -c80-250 $O/r05_bench_full_b32.json
