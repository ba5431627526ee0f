#!/bin/bash
# Round-6 evidence (GPU box, from the repo root): bench lines, per-layer tables, rocprofv3 kernel stats, PMC traffic (also for
# the training / 608 lines: VERDICT r05 item 8b) + SQ counters.   bash tools/profile_r06.sh   ->  gpurun_out/prof/ (copy into profiles/)
TAG=r06; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_$TAG.json
DATE=$(date +%Y-%m-%d)
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_full_b32.json 2> $OUT/${TAG}_bench_full_b32.err
BENCH_LAYERS=1 python bench.py --no-cpu-baseline --no-batch-sweep --steps 10 2> $OUT/${TAG}_layers_f32_b32.txt > /dev/null
BENCH_LAYERS=1 python bench.py --dtype bf16 --no-cpu-baseline --no-batch-sweep --steps 10 2> $OUT/${TAG}_layers_bf16_b32.txt > $OUT/${TAG}_bf16_bench_full_b32.json
python bench.py --workload detector --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_bench_detector_b8.json 2>/dev/null
python bench.py --workload module2 --dtype bf16 --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_bf16_bench_module2_b32.json 2>/dev/null
python bench.py --workload allreduce --steps 5 > $OUT/${TAG}_bench_allreduce_1rank_no_pg.json 2>/dev/null
BENCH_FORCE_SPAWN=1 python bench.py --workload allreduce --steps 5 > $OUT/${TAG}_bench_allreduce_rccl_world1.json 2>/dev/null
(python tools/bneck_bench.py 32; python tools/kw_bench.py 1) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_microbench.txt
for p in 0 1; do python tools/m2_train_step.py 30 8 bf16 $p 2>&1 | grep "stage-2"; done > $OUT/${TAG}_m2_train_prefetch_ab.txt
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o k -- $CMD > /tmp/kt.log 2>&1
python $R/tools/prof_summary.py /tmp/kt_$TAG/k_results.db > $OUT/${TAG}_bench_full_b32_kernel_stats.txt 2>&1
# HBM traffic: FETCH_SIZE / WRITE_SIZE need separate passes (MI355X_MICROARCH.md); one pair per measured line
traffic() {   # name  kernel-substrings  key=value...  --  bench arguments
  local name=$1 match=$2; shift 2
  local keys=(); while [ "$1" != "--" ]; do keys+=("$1"); shift; done; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$name -o f -- python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 3 --warmup 1 --prewarm-seconds 0.3 "$@" > /tmp/pf_$name.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$name -o w -- python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 3 --warmup 1 --prewarm-seconds 0.3 "$@" > /tmp/pw_$name.log 2>&1
  python $R/tools/pmc_traffic.py /tmp/pf_$name/f_results.db /tmp/pw_$name/w_results.db "$match" "${keys[@]}" date=$DATE > $OUT/$name.json
}
traffic conv_traffic conv_igemm_buf_f32 cfg=yolov3 size=416 batch=32 workload=full dtype=f32 --
traffic conv_traffic_bf16 conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws,bneck_kernel,conv_kw cfg=yolov3 size=416 batch=32 workload=full dtype=bf16 -- --dtype bf16
traffic conv_traffic_full_f16_608 conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws,bneck_kernel,conv_kw cfg=yolov3 size=608 batch=16 workload=full dtype=f16 -- --dtype f16 --size 608 --batch 16
traffic conv_traffic_train_bf16 conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws,bneck_kernel,conv_kw cfg=yolov3 size=416 batch=8 workload=train dtype=bf16 -- --workload train --dtype bf16
traffic conv_traffic_train_f32 conv_igemm_buf_f32,conv3x3_p8 cfg=yolov3 size=416 batch=8 workload=train dtype=f32 -- --workload train
traffic conv_traffic_detector_train_f32 conv_igemm_buf_f32,conv3x3_p8,conv_wgrad,wgrad9 cfg=yolov3 size=416 batch=8 workload=detector_train dtype=f32 -- --workload detector_train
traffic conv_traffic_detector_train_bf16 conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws,conv_wgrad cfg=yolov3 size=416 batch=8 workload=detector_train dtype=bf16 -- --workload detector_train --dtype bf16
traffic conv_traffic_detector_train_bf16_graph conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws,conv_wgrad cfg=yolov3 size=416 batch=8 workload=detector_train dtype=bf16 -- --workload detector_train --dtype bf16 --graph
python $R/tools/pmc_layers.py /tmp/pf_conv_traffic/f_results.db /tmp/pw_conv_traffic/w_results.db 32 416 f32 > $OUT/${TAG}_layer_traffic_f32.txt 2>&1
python $R/tools/pmc_layers.py /tmp/pf_conv_traffic_bf16/f_results.db /tmp/pw_conv_traffic_bf16/w_results.db 32 416 bf16 > $OUT/${TAG}_layer_traffic_bf16.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace -d /tmp/pms_$TAG -o s -- $CMD > /tmp/pms.log 2>&1
python $R/tools/prof_summary.py /tmp/pms_$TAG/s_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_sq.txt 2>&1
cd $R
# the lines whose roofline.traffic comes from the passes above (the json files must be in profiles/ for bench.py to find them)
cp $OUT/conv_traffic*.json $R/profiles/
python bench.py --workload train --no-cpu-baseline --steps 20 > $OUT/${TAG}_bench_train_b8.json 2>/dev/null
python bench.py --workload train --dtype bf16 --no-cpu-baseline --steps 20 > $OUT/${TAG}_bf16_bench_train_b8.json 2>/dev/null
python bench.py --workload train --dtype bf16 --no-prefetch --no-cpu-baseline --steps 20 > $OUT/${TAG}_bf16_bench_train_b8_no_prefetch.json 2>/dev/null
python bench.py --workload train --no-prefetch --no-cpu-baseline --steps 20 > $OUT/${TAG}_bench_train_b8_no_prefetch.json 2>/dev/null
python bench.py --dtype f16 --size 608 --batch 16 --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_f16_bench_full_608_b16.json 2>/dev/null
python bench.py --workload detector_train --no-cpu-baseline --steps 20 --warmup 3 > $OUT/${TAG}_bench_detector_train_b8.json 2>/dev/null
python bench.py --workload detector_train --dtype bf16 --no-cpu-baseline --steps 20 --warmup 3 > $OUT/${TAG}_bf16_bench_detector_train_b8.json 2>/dev/null
python bench.py --workload detector_train --dtype bf16 --graph --no-cpu-baseline --steps 20 --warmup 3 > $OUT/${TAG}_bf16_bench_detector_train_b8_graph.json 2>/dev/null
cat $OUT/conv_traffic*.json | cut -c1-300
tail -n 2 $OUT/${TAG}_layer_traffic_f32.txt $OUT/${TAG}_layer_traffic_bf16.txt
ls -la $OUT
