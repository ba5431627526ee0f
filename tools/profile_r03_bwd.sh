#!/bin/bash
# Round-3 evidence for the detector backward (row a6): bench line + rocprofv3 kernel stats of the fp32 training step at batch 8.
# Two profiled runs (8 and 28 steps): their difference per training step (= per whole-network pack launch) is the steady-state step (tools/prof_diff.py) - planning, autotuning and
# the accuracy passes of the process cancel.
TAG=${TAG:-r03}; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_$TAG.json
python bench.py --workload detector_train --no-cpu-baseline --steps 10 --warmup 3 > $OUT/${TAG}_bench_detector_train_b8.json 2> $OUT/${TAG}_bench_detector_train_b8.err
cd /tmp   # (the bench run above measured and cached the training path's conv tiles: the profiled runs are steady state)
rocprofv3 --kernel-trace --stats -d /tmp/ktb8_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 6 --warmup 2 > /tmp/ktb8.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ktb_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 26 --warmup 2 > /tmp/ktb.log 2>&1
python $R/tools/prof_summary.py /tmp/ktb_$TAG/k_results.db > $OUT/${TAG}_bench_detector_train_b8_kernel_stats.txt 2>&1
python $R/tools/prof_summary.py /tmp/ktb_$TAG/k_results.db --by-grid > $OUT/${TAG}_bench_detector_train_b8_by_grid.txt 2>&1
(cd $R/tools; python prof_diff.py /tmp/ktb8_$TAG/k_results.db /tmp/ktb_$TAG/k_results.db pack_conv_batch_kernel) > $OUT/${TAG}_bench_detector_train_b8_per_step_overlap.txt 2>&1
# the same with the weight gradients on the main stream: a kernel's duration is its own (the per-kernel rooflines of DESIGN section 3)
MILLIEYE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/kts8_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 6 --warmup 2 > /tmp/kts8.log 2>&1
MILLIEYE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/kts_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 26 --warmup 2 > /tmp/kts.log 2>&1
(cd $R/tools; python prof_diff.py /tmp/kts8_$TAG/k_results.db /tmp/kts_$TAG/k_results.db pack_conv_batch_kernel) > $OUT/${TAG}_bench_detector_train_b8_per_step.txt 2>&1
cat $OUT/${TAG}_bench_detector_train_b8.json | cut -c1-300
head -45 $OUT/${TAG}_bench_detector_train_b8_per_step.txt; head -8 $OUT/${TAG}_bench_detector_train_b8_per_step_overlap.txt
