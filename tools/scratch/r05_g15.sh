cd /root/repo; mkdir -p gpurun_out/r05o; O=gpurun_out/r05o
timeout 900 python -m pytest tests/test_gpu_train16.py -q -m gpu -s 2>&1 | tail -15 > $O/tests16.txt
timeout 900 python tools/train16_check.py bf16 8 416 2>&1 | grep -v "amdgpu\|Warning\|detach\|print(" > $O/train16_check_bf16.txt
timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_bf16.json 2> $O/bench_detector_train_bf16.err
MILLIEYE_WGRAD16=0 timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_bf16_w32.json 2>/dev/null
timeout 900 python bench.py --workload detector_train --dtype f16 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_f16.json 2>/dev/null
cat $O/tests16.txt $O/train16_check_bf16.txt; cut -c80-200 $O/bench_detector_train_bf16.json $O/bench_detector_train_bf16_w32.json $O/bench_detector_train_f16.json
