cd /root/repo; mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/chain_probe.hip && CHAIN_DEBUG=1 timeout 60 /tmp/chain_probe > $O/chain_probe_dbg.txt 2>&1; head -30 $O/chain_probe_dbg.txt
timeout 240 /tmp/chain_probe > $O/chain_probe.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_darknet.py -x -q -m gpu -k "tap_masks or backward or train or wgrad or affine" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
for w in 256 512 1024; do echo "AFFINE_ROWS=1 WGS=$w" >> $O/affine_bench.txt; MILLIEYE_AFFINE_ROWS=1 MILLIEYE_AFFINE_WGS=$w timeout 300 python tools/affine_bench.py 8 2>&1 | grep -v amdgpu >> $O/affine_bench.txt; done
echo "masks on" >> $O/dgrad_bench.txt; timeout 300 python tools/dgrad_bench.py 8 2>&1 | grep "s2\|total" >> $O/dgrad_bench.txt
echo "masks off" >> $O/dgrad_bench.txt; MILLIEYE_PARITY_MASKS=0 timeout 300 python tools/dgrad_bench.py 8 2>&1 | grep "s2\|total" >> $O/dgrad_bench.txt
timeout 600 python bench.py --workload detector_train --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train.json 2> $O/bench_detector_train.err
MILLIEYE_PARITY_MASKS=0 MILLIEYE_WGRAD_DEPTH=1 timeout 600 python bench.py --workload detector_train --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_old.json 2> $O/bench_detector_train_old.err
timeout 900 python bench.py --no-cpu-baseline --no-accuracy > $O/bench_default.json 2> $O/bench_default.err
cat $O/chain_probe.txt; tail -3 $O/tests.txt; cat $O/affine_bench.txt | grep "total\|AFF"; cat $O/dgrad_bench.txt; cut -c1-200 $O/bench_detector_train.json $O/bench_detector_train_old.json; cut -c1-300 $O/bench_default.json
