#!/bin/bash
# Refresh the evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r01     ->  gpurun_out/prof/<tag>_*.{json,txt}   (copy them into profiles/)
# 1. default bench line (+ cpu_baseline) and the detector / train workloads
# 2. rocprofv3 --kernel-trace --stats of the default command (summary via tools/prof_summary.py)
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE - never combined with trace domains other than kernel-trace)
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_$TAG.json     # first bench run tunes; the profiled runs reuse it
python bench.py > $OUT/${TAG}_bench_full_b32.json 2> $OUT/bench_full.err
python bench.py --workload detector --no-cpu-baseline > $OUT/${TAG}_bench_detector_b8.json 2>> $OUT/bench_full.err
python bench.py --workload train --no-cpu-baseline > $OUT/${TAG}_bench_train_b8.json 2>> $OUT/bench_full.err
python bench.py --workload detector_train --no-cpu-baseline > $OUT/${TAG}_bench_detector_train_b8.json 2>> $OUT/bench_full.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pf_$TAG -o full -- python $R/bench.py --no-cpu-baseline > /tmp/pf.log 2>&1
python $R/tools/prof_summary.py /tmp/pf_$TAG/full_results.db | head -40 > $OUT/${TAG}_bench_full_b32_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf_$TAG -o f -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pmf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw_$TAG -o w -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pmw.log 2>&1
python $R/tools/prof_summary.py /tmp/pmf_$TAG/f_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_fetch.txt
python $R/tools/prof_summary.py /tmp/pmw_$TAG/w_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_write.txt
python $R/tools/pmc_traffic.py /tmp/pmf_$TAG/f_results.db /tmp/pmw_$TAG/w_results.db conv_igemm_buf_f32 > $OUT/conv_traffic.json
python $R/tools/pmc_traffic.py /tmp/pmf_$TAG/f_results.db /tmp/pmw_$TAG/w_results.db conv_igemm_buf_h16 > $OUT/conv_traffic_bf16.json
cat $OUT/${TAG}_bench_full_b32.json
