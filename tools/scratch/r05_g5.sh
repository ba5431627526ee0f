cd /root/repo; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_darknet.py -x -q -m gpu -k "backward or train or affine" > $O/tests_train.txt 2>&1; echo "tests rc=$?" >> $O/tests_train.txt
timeout 300 python tools/affine_bench.py 8 > $O/affine_bench.txt 2>&1
timeout 300 python tools/wgrad_bench.py 8 > $O/wgrad_bench.txt 2>&1
timeout 300 python tools/dgrad_bench.py 8 > $O/dgrad_bench.txt 2>&1
for b in 1 8; do for dt in f32 bf16; do timeout 300 python tools/b1_tail_events.py $b $dt >> $O/b1_tail_events.txt 2>&1; done; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/chain_probe.hip && timeout 300 /tmp/chain_probe > $O/chain_probe.txt 2>&1
for st in 0 300 600 1000 1500 2500; do echo "STAGGER=$st" >> $O/p8_stagger.txt; MILLIEYE_P8_STAGGER=$st timeout 300 python tools/p8_bench.py 32 200,201,221,621 2>&1 | grep -v amdgpu | cut -c1-260 >> $O/p8_stagger.txt; done
timeout 600 python tools/bf16_row_diff.py bf16 > $O/bf16_row_diff.txt 2>&1
tail -3 $O/tests_train.txt; cat $O/affine_bench.txt; cat $O/chain_probe.txt; cat $O/p8_stagger.txt; cat $O/b1_tail_events.txt
