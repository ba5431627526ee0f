#!/bin/bash
# Evidence for the 16-bit storage modes (bf16; f16 where noted) (run on the GPU box from the repo root):
#   bash tools/profile_bf16.sh r01   ->  gpurun_out/prof/<tag>_bf16_*   (copy into profiles/)
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune16_$TAG.json
python bench.py --dtype bf16 --no-cpu-baseline > $OUT/${TAG}_bf16_bench_full_b32.json 2> $OUT/bench_bf16.err
python bench.py --dtype bf16 --no-cpu-baseline --workload detector > $OUT/${TAG}_bf16_bench_detector_b8.json 2>> $OUT/bench_bf16.err
python bench.py --dtype bf16 --no-cpu-baseline --size 608 --batch 16 > $OUT/${TAG}_bf16_bench_full_608_b16.json 2>> $OUT/bench_bf16.err
{
  for b in 1 8 32 64; do
    python bench.py --dtype bf16 --no-cpu-baseline --batch $b 2>> $OUT/bench_bf16.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 full  batch %3d: %8.1f frames/s  %7.3f ms/step  conv %.0f TF' % ($b, d['value'], d['ms_per_step'], d['roofline']['achieved']))"
    python bench.py --dtype bf16 --no-cpu-baseline --workload detector --batch $b 2>> $OUT/bench_bf16.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 detector batch %3d: %8.1f frames/s  %7.3f ms/step  conv %.0f TF' % ($b, d['value'], d['ms_per_step'], d['roofline']['achieved']))"
  done
} > $OUT/${TAG}_bf16_batch_sweep.txt
python tools/conv16_bench.py 32 > $OUT/${TAG}_bf16_conv_bench_b32.txt 2>&1
python tools/conv16_bench.py 32 13,14,73,74,83,84,93,94 > $OUT/${TAG}_bf16_conv_ablation_b32.txt 2>&1
python tools/bf16_error_stats.py bf16 > $OUT/${TAG}_bf16_error_stats.txt 2>&1
python tools/bf16_error_stats.py f16 > $OUT/${TAG}_f16_error_stats.txt 2>&1
# IEEE-half storage (BASELINE configs[4]: "608x608 input, fp16 MFMA convs", 16 frames per GPU) and the default shape
python bench.py --dtype f16 --no-cpu-baseline --size 608 --batch 16 > $OUT/${TAG}_f16_bench_full_608_b16.json 2>> $OUT/bench_bf16.err
python bench.py --dtype f16 --no-cpu-baseline > $OUT/${TAG}_f16_bench_full_b32.json 2>> $OUT/bench_bf16.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pf16_$TAG -o full -- python $R/bench.py --dtype bf16 --no-cpu-baseline > /tmp/pf16.log 2>&1
python $R/tools/prof_summary.py /tmp/pf16_$TAG/full_results.db | head -40 > $OUT/${TAG}_bf16_bench_full_b32_kernel_stats.txt
cat $OUT/${TAG}_bf16_bench_full_b32.json | head -c 600; echo; cat $OUT/${TAG}_bf16_batch_sweep.txt
