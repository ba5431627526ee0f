cd /root/repo; mkdir -p gpurun_out/r05l; O=gpurun_out/r05l
timeout 1200 python -m pytest tests/test_gpu_train16.py -x -q -m gpu -s > $O/tests16.txt 2>&1; echo "tests rc=$?" >> $O/tests16.txt
timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_bf16.json 2> $O/bench_detector_train_bf16.err
timeout 600 python bench.py --workload detector_train --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_f32.json 2> $O/bench_detector_train_f32.err
tail -25 $O/tests16.txt; cut -c1-700 $O/bench_detector_train_bf16.json; tail -5 $O/bench_detector_train_bf16.err; cut -c80-200 $O/bench_detector_train_f32.json
