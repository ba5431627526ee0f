set -x
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tail_split or buffer_addressed or split_k" 2>&1 | tail -5
for t in 1 2 3 5; do
timeout 600 python tools/conv_bench.py --batch 32 --tiles $t,4$t --splits 1,2,3,4,6 --reps 10 --prewarm 0.15 2>&1 | grep -v "^$" | tail -20
done
