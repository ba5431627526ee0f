cd /root/repo; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/chain_probe.hip && timeout 240 /tmp/chain_probe > $O/chain_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_darknet.py tests/test_gpu_configs.py -x -q -m gpu -k "tap_masks or backward or 16bit" -s > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
timeout 600 python tools/bf16_row_diff.py bf16 > $O/bf16_row_diff.txt 2>&1
timeout 300 python tools/dgrad_bench.py 8 2>&1 | grep "s2\|total" > $O/dgrad_bench.txt
for i in 1 2; do
timeout 600 python bench.py --workload detector_train --batch 8 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $O/detector_train_ab.txt
MILLIEYE_PARITY_MASKS=0 MILLIEYE_WGRAD_DEPTH=1 timeout 600 python bench.py --workload detector_train --batch 8 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $O/detector_train_ab.txt
done
cat $O/chain_probe.txt; tail -3 $O/tests.txt; grep "m2b32" $O/tests.txt; grep "==\|tie-aware" $O/bf16_row_diff.txt; cat $O/dgrad_bench.txt $O/detector_train_ab.txt
