cd /root/repo; mkdir -p gpurun_out/r05s; O=gpurun_out/r05s
timeout 1500 python -m pytest tests/test_gpu_train16.py tests/test_gpu_network.py tests/test_gpu_parallel.py -x -q -m gpu -k "16bit or train16 or h16 or pooled or two_ranks_inference" 2>&1 | tail -6 > $O/tests.txt
timeout 900 python tools/train16_check.py bf16 8 416 2>&1 | grep "batch 8\|step 25" > $O/train16_check.txt
for i in 1 2; do timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c80-170 >> $O/bench16.txt; done
timeout 300 python tools/detector_train16_host_profile.py bf16 8 2>&1 | grep "host issue" >> $O/bench16.txt
cat $O/tests.txt $O/train16_check.txt $O/bench16.txt
