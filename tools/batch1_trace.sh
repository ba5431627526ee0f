#!/bin/bash
# Batch-1 / batch-8 latency evidence (GPU box): bench line + the dispatch timeline of the last forward (duration and idle gap per launch).
#   bash tools/batch1_trace.sh [TAG]    -> gpurun_out/prof/${TAG}_b{1,8}_{f32,bf16}_{bench.json,timeline.txt}
TAG=${1:-r04}; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_b1_$TAG.json
for b in 1 8; do
  for dt in f32 bf16; do
    A="--workload full --batch $b --dtype $dt --no-cpu-baseline --no-batch-sweep --no-bf16-line --no-accuracy"
    python bench.py $A --steps 50 --warmup 10 > $OUT/${TAG}_b${b}_${dt}_bench.json 2>/dev/null
    (cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/kt_b${b}_${dt} -o k -- python $R/bench.py $A --steps 3 --warmup 2 > /tmp/kt_b1.log 2>&1)
    python tools/prof_summary.py /tmp/kt_b${b}_${dt}/k_results.db --timeline 140 > $OUT/${TAG}_b${b}_${dt}_timeline.txt 2>&1
  done
done
