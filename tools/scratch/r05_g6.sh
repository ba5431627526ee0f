cd /root/repo; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" >> $O/gpu_tests.txt
for v in 0 1; do echo "AFFINE_ROWS=$v" >> $O/affine_bench.txt; MILLIEYE_AFFINE_ROWS=$v timeout 300 python tools/affine_bench.py 8 2>&1 | grep -v amdgpu >> $O/affine_bench.txt; done
echo "DEPTH=1" >> $O/wgrad_bench.txt; MILLIEYE_WGRAD_DEPTH=1 timeout 300 python tools/wgrad_bench.py 8 2>&1 | grep -v amdgpu >> $O/wgrad_bench.txt
echo "DEPTH=4" >> $O/wgrad_bench.txt; timeout 300 python tools/wgrad_bench.py 8 2>&1 | grep -v amdgpu >> $O/wgrad_bench.txt
for cfg in "128 1024 256" "128 512 256" "128 512 128" "64 1024 256" "64 4096 128" "128 2048 128"; do set -- $cfg; echo "DEPTH=4 TILE=$1 WGS=$2 MINPX=$3" >> $O/wgrad_sweep.txt; MILLIEYE_WGRAD_TILE=$1 MILLIEYE_WGRAD_WGS=$2 MILLIEYE_WGRAD_MINPX=$3 timeout 300 python tools/wgrad_bench.py 8 2>&1 | grep "k1 s1\|total" >> $O/wgrad_sweep.txt; done
for dt in f32 bf16; do timeout 300 python tools/b1_tail_events.py 1 $dt 2>&1 | grep -v amdgpu >> $O/b1_tail_events.txt; done
for dt in f32 bf16; do MILLIEYE_COUNT_SPIN=0 timeout 300 python tools/b1_tail_events.py 1 $dt 2>&1 | grep "ms/step" >> $O/b1_tail_events_nospin.txt; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/chain_probe.hip && timeout 200 /tmp/chain_probe > $O/chain_probe.txt 2>&1
timeout 600 python bench.py --workload detector_train --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train.json 2> $O/bench_detector_train.err
MILLIEYE_AFFINE_ROWS=0 MILLIEYE_WGRAD_DEPTH=1 timeout 600 python bench.py --workload detector_train --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_detector_train_old.json 2> $O/bench_detector_train_old.err
timeout 600 python tools/bf16_row_diff.py bf16 > $O/bf16_row_diff.txt 2>&1
tail -3 $O/gpu_tests.txt; cat $O/affine_bench.txt; cat $O/wgrad_bench.txt; cat $O/chain_probe.txt; cat $O/b1_tail_events.txt $O/b1_tail_events_nospin.txt; cut -c1-300 $O/bench_detector_train.json $O/bench_detector_train_old.json
