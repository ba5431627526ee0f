#!/bin/bash
# Round-6 evidence, second pass: the two batch-32 inference lines' counter traffic from the LAST forward of each PMC pass
# (pmc_traffic.py last_forward=74; MILLIEYE_BNECK=0 so that the launch list is the 74 conv launches on every box - under the profiler
# the measured pair-vs-one-launch choice can flip), the per-layer tables from the same passes, then the inference lines again so
# that their roofline.traffic is filled from the files of THIS kernel generation.
TAG=r06; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_${TAG}b.json
DATE=$(date +%Y-%m-%d)
python bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 3 --warmup 1 > /dev/null 2>&1   # tune both engines
cd /tmp
C32="python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 3 --warmup 1 --prewarm-seconds 0.3"
for dt in f32 bf16; do
  EXTRA=""; [ $dt = bf16 ] && EXTRA="--dtype bf16"
  MILLIEYE_BNECK=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$dt -o f -- $C32 $EXTRA > /tmp/pf_$dt.log 2>&1
  MILLIEYE_BNECK=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$dt -o w -- $C32 $EXTRA > /tmp/pw_$dt.log 2>&1
  NAME=conv_traffic; [ $dt = bf16 ] && NAME=conv_traffic_bf16
  python $R/tools/pmc_traffic.py /tmp/pf_$dt/f_results.db /tmp/pw_$dt/w_results.db conv cfg=yolov3 size=416 batch=32 workload=full dtype=$dt last_forward=74 date=$DATE > $OUT/$NAME.json
  python $R/tools/pmc_layers.py /tmp/pf_$dt/f_results.db /tmp/pw_$dt/w_results.db 32 416 $dt > $OUT/${TAG}_layer_traffic_$dt.txt 2>&1
  tail -n 2 $OUT/${TAG}_layer_traffic_$dt.txt
done
cd $R
cp $OUT/conv_traffic.json $OUT/conv_traffic_bf16.json $R/profiles/
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_full_b32.json 2> $OUT/${TAG}_bench_full_b32.err
python bench.py --dtype bf16 --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_bf16_bench_full_b32.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06_bench_full_b32.json", "r06_bf16_bench_full_b32.json"):
    d = json.loads(open("gpurun_out/prof/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], (d.get("bf16_storage_mode") or {}).get("value"),
          ((d.get("bf16_storage_mode") or {}).get("roofline") or {}).get("traffic"))
PY
# second half of the round: micro-benchmarks and per-step kernel tables of the two training steps
(for e in 0 1; do MILLIEYE_ROI_BWD_LDS=$e python tools/roi_bwd_bench.py 1200 8; done
 for s in 4 8 16 32 64; do echo "slices per frame: $s"; MILLIEYE_ROI_BWD_SPLITS=$s python tools/roi_bwd_bench.py 1200 8 | tail -n 1; done
 MILLIEYE_ROI_BWD_LDS=0 python tools/roi_bwd_bench.py 200 1; python tools/roi_bwd_bench.py 200 1) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_micro_roi_bwd_lds.txt
(python tools/linear_bench.py; MILLIEYE_M2_LINEAR_NAIVE=1 python tools/linear_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_micro_linear.txt
python tools/memset_graph_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_memset_graph_probe.txt
(for m in philox cpu; do for p in 1 0; do M2_DROPOUT=$m python tools/m2_train_step.py 40 8 bf16 $p 2>&1 | grep "stage-2" | sed "s/\$/  [mask: $m]/"; done; done) > $OUT/${TAG}_m2_train_prefetch_ab.txt
cd /tmp
TR="python $R/bench.py --workload train --dtype bf16 --no-cpu-baseline --warmup 4"
rocprofv3 --kernel-trace --stats -d /tmp/ts_a -o k -- $TR --steps 10 > /tmp/ts_a.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ts_b -o k -- $TR --steps 50 > /tmp/ts_b.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/m2_a -o k -- python $R/tools/m2_train_step.py 10 8 bf16 1 > /tmp/m2_a.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/m2_b -o k -- python $R/tools/m2_train_step.py 40 8 bf16 1 > /tmp/m2_b.log 2>&1
cd $R/tools
python prof_diff.py /tmp/ts_a/k_results.db /tmp/ts_b/k_results.db heads_tail_bwd 2>&1 | head -n 70 > $OUT/${TAG}_train_step_per_step_kernels.txt
python prof_diff.py /tmp/m2_a/k_results.db /tmp/m2_b/k_results.db 30 2>&1 | head -n 70 > $OUT/${TAG}_m2_train_step_per_step_kernels.txt
cd $R
head -n 3 $OUT/${TAG}_train_step_per_step_kernels.txt $OUT/${TAG}_m2_train_step_per_step_kernels.txt; cat $OUT/${TAG}_m2_train_prefetch_ab.txt
