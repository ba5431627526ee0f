cd /root/repo; mkdir -p gpurun_out/r05w; O=gpurun_out/r05w
timeout 900 python -m pytest tests/test_gpu_h16.py -x -q -m gpu -k "bf16_tiles" 2>&1 | tail -4 > $O/tests.txt
timeout 900 python tools/conv16_bench.py 32 2>&1 | grep "s2" | cut -c1-250 > $O/conv16_bench_s2.txt
for i in 1 2; do
MILLIEYE_NO_KORD16=1 MILLIEYE_TUNE_CACHE=/tmp/tune_a.json timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-accuracy --no-batch-sweep 2>/dev/null | cut -c80-170 >> $O/bench_ab.txt
MILLIEYE_TUNE_CACHE=/tmp/tune_b.json timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-accuracy --no-batch-sweep 2>/dev/null | cut -c80-170 >> $O/bench_ab.txt
done
cat $O/tests.txt $O/conv16_bench_s2.txt $O/bench_ab.txt
