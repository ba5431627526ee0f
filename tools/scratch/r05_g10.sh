cd /root/repo; mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
timeout 600 python -m pytest tests/test_gpu_h16.py -x -q -m gpu -k "bf16_tiles" > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
timeout 900 python tools/conv16_bench.py 32 2>&1 | grep -v amdgpu | cut -c1-260 > $O/conv16_bench_b32.txt
timeout 600 python tools/conv16_bench.py 1 2>&1 | grep -v amdgpu | cut -c1-260 > $O/conv16_bench_b1.txt
MILLIEYE_NO_KSUB4=1 timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-accuracy 2>/dev/null | cut -c1-200 > $O/bench_bf16_ab.txt
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-accuracy 2>/dev/null | cut -c1-200 >> $O/bench_bf16_ab.txt
MILLIEYE_NO_KSUB4=1 timeout 600 python bench.py --dtype bf16 --batch 1 --no-cpu-baseline --no-accuracy --steps 200 2>/dev/null | cut -c1-200 >> $O/bench_bf16_ab.txt
timeout 600 python bench.py --dtype bf16 --batch 1 --no-cpu-baseline --no-accuracy --steps 200 2>/dev/null | cut -c1-200 >> $O/bench_bf16_ab.txt
tail -3 $O/tests.txt; cat $O/conv16_bench_b32.txt $O/conv16_bench_b1.txt $O/bench_bf16_ab.txt
