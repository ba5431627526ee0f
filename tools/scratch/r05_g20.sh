cd /root/repo; mkdir -p gpurun_out/r05t; O=gpurun_out/r05t
timeout 600 python -m pytest tests/test_gpu_network.py -x -q -m gpu -k "pooled" 2>&1 | tail -4 > $O/tests.txt
for i in 1 2 3; do
timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c80-170 >> $O/bench16.txt
MILLIEYE_WGRAD16_DIRECT=0 timeout 900 python bench.py --workload detector_train --dtype bf16 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c80-170 >> $O/bench16.txt
done
cat $O/tests.txt $O/bench16.txt
