#!/bin/bash
# End-of-round validation on a GPU box (gpurun -- 'bash tools/validate_final.sh'): the GPU suite, the bench lines the docs quote, the
# PMC traffic passes on the sources as they are (profiles/conv_traffic*.json must carry bench.kernel_generation()), the captured
# training step's kernel statistics.  Results under gpurun_out/final/; copy what is to be kept into profiles/.
cd "$(dirname "$0")/.."; R=$PWD; OUT=$R/gpurun_out/final; mkdir -p $OUT; rm -f $OUT/*; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_final.json
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $OUT/tests.txt
python bench.py --workload detector_train --no-cpu-baseline --steps 20 --warmup 3 > $OUT/r05_bench_detector_train_b8.json 2>/dev/null
python bench.py --workload detector_train --dtype bf16 --no-cpu-baseline --steps 20 --warmup 3 > $OUT/r05_bf16_bench_detector_train_b8.json 2>/dev/null
python bench.py --workload detector_train --dtype bf16 --graph --no-cpu-baseline --steps 20 --warmup 3 > $OUT/r05_bf16_bench_detector_train_b8_graph.json 2>/dev/null
python bench.py --workload detector_train --dtype f16 --graph --no-cpu-baseline --steps 20 --warmup 3 > $OUT/r05_f16_bench_detector_train_b8_graph.json 2>/dev/null
python bench.py --workload train --no-cpu-baseline --steps 20 > $OUT/r05_bench_train_b8.json 2>/dev/null
python bench.py --workload train --dtype bf16 --no-cpu-baseline --steps 20 > $OUT/r05_bf16_bench_train_b8.json 2>/dev/null
python bench.py --workload train --no-prefetch --no-cpu-baseline --steps 20 > $OUT/r05_bench_train_b8_no_prefetch.json 2>/dev/null
python bench.py --workload train --dtype bf16 --no-prefetch --no-cpu-baseline --steps 20 > $OUT/r05_bf16_bench_train_b8_no_prefetch.json 2>/dev/null
(python tools/wgrad_bench.py 8 bf16; python tools/affine_bench.py 8 bf16) 2>&1 | grep -v amdgpu > $OUT/r05_backward_microbench_bf16.txt
DATE=$(date +%Y-%m-%d)
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 3 --warmup 1"
C32="$CMD --no-bf16-line --prewarm-seconds 0.3"
C16="$C32 --dtype bf16"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf -o f -- $C32 > /tmp/pmf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw -o w -- $C32 > /tmp/pmw.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf16 -o f -- $C16 > /tmp/pmf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw16 -o w -- $C16 > /tmp/pmw16.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmf/f_results.db /tmp/pmw/w_results.db conv_igemm_buf_f32 cfg=yolov3 size=416 batch=32 workload=full dtype=f32 date=$DATE > $OUT/conv_traffic.json
python $R/tools/pmc_traffic.py /tmp/pmf16/f_results.db /tmp/pmw16/w_results.db conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws cfg=yolov3 size=416 batch=32 workload=full dtype=bf16 date=$DATE > $OUT/conv_traffic_bf16.json
rocprofv3 --kernel-trace --stats -d /tmp/ktg -o k -- python $R/bench.py --workload detector_train --dtype bf16 --graph --no-cpu-baseline --steps 26 --warmup 2 > /tmp/ktg.log 2>&1
python $R/tools/prof_summary.py /tmp/ktg/k_results.db > $OUT/r05_bf16_bench_detector_train_b8_graph_kernel_stats.txt 2>&1
python $R/tools/step_overlap.py /tmp/ktg/k_results.db > $OUT/r05_bf16_bench_detector_train_b8_graph_overlap.txt 2>&1
cd $R
cp $OUT/conv_traffic.json $OUT/conv_traffic_bf16.json profiles/   # (on the box: the default bench line below reads them)
python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_full_b32.json 2> $OUT/r05_bench_full_b32.err
tail -3 $OUT/tests.txt
for f in $OUT/r05_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('traffic'), (d.get('roofline') or {}).get('frac'), (d.get('bf16_storage_mode') or {}).get('value'))"; done
