#!/bin/bash
# Round-5 evidence (GPU box): bench lines, per-layer tables, rocprofv3 kernel stats, PMC traffic + SQ counters.
TAG=r05; R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
export MILLIEYE_TUNE_CACHE=/tmp/tune_$TAG.json
DATE=$(date +%Y-%m-%d)
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_full_b32.json 2> $OUT/${TAG}_bench_full_b32.err
BENCH_LAYERS=1 python bench.py --no-cpu-baseline --no-batch-sweep --steps 10 2> $OUT/${TAG}_layers_f32_b32.txt > /dev/null
BENCH_LAYERS=1 python bench.py --dtype bf16 --no-cpu-baseline --no-batch-sweep --steps 10 2> $OUT/${TAG}_layers_bf16_b32.txt > $OUT/${TAG}_bf16_bench_full_b32.json
python bench.py --workload detector --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_bench_detector_b8.json 2>/dev/null
python bench.py --workload train --no-cpu-baseline --steps 20 > $OUT/${TAG}_bench_train_b8.json 2>/dev/null
python bench.py --workload train --dtype bf16 --no-cpu-baseline --steps 20 > $OUT/${TAG}_bf16_bench_train_b8.json 2>/dev/null
python bench.py --dtype f16 --size 608 --batch 16 --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_f16_bench_full_608_b16.json 2>/dev/null
python bench.py --workload module2 --dtype bf16 --no-cpu-baseline --no-batch-sweep --steps 20 > $OUT/${TAG}_bf16_bench_module2_b32.json 2>/dev/null
python bench.py --workload detector_train --no-cpu-baseline --steps 20 --warmup 3 > $OUT/${TAG}_bench_detector_train_b8.json 2>/dev/null
python bench.py --workload detector_train --dtype bf16 --no-cpu-baseline --steps 20 --warmup 3 > $OUT/${TAG}_bf16_bench_detector_train_b8.json 2>/dev/null
python bench.py --workload detector_train --dtype bf16 --graph --no-cpu-baseline --steps 20 --warmup 3 > $OUT/${TAG}_bf16_bench_detector_train_b8_graph.json 2>/dev/null
python bench.py --workload allreduce --steps 5 > $OUT/${TAG}_bench_allreduce_1rank_no_pg.json 2>/dev/null
BENCH_FORCE_SPAWN=1 python bench.py --workload allreduce --steps 5 > $OUT/${TAG}_bench_allreduce_rccl_world1.json 2>/dev/null
(python tools/affine_bench.py 8; python tools/conv16_bench.py 32; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/chain_probe.hip && timeout 300 /tmp/chain_probe) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_microbench.txt
for b in 1 8; do for dt in f32 bf16; do python tools/b1_tail_events.py $b $dt 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_tail_events.txt; done; done
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o k -- $CMD > /tmp/kt.log 2>&1
python $R/tools/prof_summary.py /tmp/kt_$TAG/k_results.db > $OUT/${TAG}_bench_full_b32_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ktb8_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 6 --warmup 2 > /tmp/ktb8.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ktb_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 26 --warmup 2 > /tmp/ktb.log 2>&1
python $R/tools/prof_summary.py /tmp/ktb_$TAG/k_results.db > $OUT/${TAG}_bench_detector_train_b8_kernel_stats.txt 2>&1
python $R/tools/prof_summary.py /tmp/ktb_$TAG/k_results.db --by-grid > $OUT/${TAG}_bench_detector_train_b8_by_grid.txt 2>&1
(cd $R/tools; python prof_diff.py /tmp/ktb8_$TAG/k_results.db /tmp/ktb_$TAG/k_results.db pack_conv_batch_kernel) > $OUT/${TAG}_bench_detector_train_b8_per_step_overlap.txt 2>&1
# the same with the weight gradients on the main stream: a kernel's duration is its own (the per-kernel rooflines of DESIGN section 3)
MILLIEYE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/kts8_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 6 --warmup 2 > /tmp/kts8.log 2>&1
MILLIEYE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/kts_$TAG -o k -- python $R/bench.py --workload detector_train --no-cpu-baseline --steps 26 --warmup 2 > /tmp/kts.log 2>&1
(cd $R/tools; python prof_diff.py /tmp/kts8_$TAG/k_results.db /tmp/kts_$TAG/k_results.db pack_conv_batch_kernel) > $OUT/${TAG}_bench_detector_train_b8_per_step.txt 2>&1
(cd $R; python tools/wgrad_bench.py 8; python tools/dgrad_bench.py 8; python tools/pack_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_backward_microbench.txt
# HBM traffic: FETCH_SIZE / WRITE_SIZE need separate passes; one pair per storage mode, every forward of the pass at batch 32
C32="$CMD --no-bf16-line --prewarm-seconds 0.3"
C16="$C32 --dtype bf16"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf_$TAG -o f -- $C32 > /tmp/pmf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw_$TAG -o w -- $C32 > /tmp/pmw.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmf16_$TAG -o f -- $C16 > /tmp/pmf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmw16_$TAG -o w -- $C16 > /tmp/pmw16.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pmf_$TAG/f_results.db /tmp/pmw_$TAG/w_results.db conv_igemm_buf_f32 cfg=yolov3 size=416 batch=32 workload=full dtype=f32 date=$DATE > $OUT/conv_traffic.json
python $R/tools/pmc_traffic.py /tmp/pmf16_$TAG/f_results.db /tmp/pmw16_$TAG/w_results.db conv_igemm_buf_h16,conv3x3_p8,conv1x1_ws,conv3x3_ws cfg=yolov3 size=416 batch=32 workload=full dtype=bf16 date=$DATE > $OUT/conv_traffic_bf16.json
python $R/tools/pmc_layers.py /tmp/pmf_$TAG/f_results.db /tmp/pmw_$TAG/w_results.db 32 416 f32 > $OUT/${TAG}_layer_traffic_f32.txt 2>&1
python $R/tools/pmc_layers.py /tmp/pmf16_$TAG/f_results.db /tmp/pmw16_$TAG/w_results.db 32 416 bf16 > $OUT/${TAG}_layer_traffic_bf16.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace -d /tmp/pms_$TAG -o s -- $CMD > /tmp/pms.log 2>&1
python $R/tools/prof_summary.py /tmp/pms_$TAG/s_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_sq.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d /tmp/pmg_$TAG -o g -- $CMD > /tmp/pmg.log 2>&1
python $R/tools/prof_summary.py /tmp/pmg_$TAG/g_results.db --pmc | grep -v "at::\|rocprim\|rocclr" > $OUT/${TAG}_bench_full_b32_pmc_grbm.txt 2>&1
cat $OUT/conv_traffic.json $OUT/conv_traffic_bf16.json
tail -n 2 $OUT/${TAG}_layer_traffic_f32.txt $OUT/${TAG}_layer_traffic_bf16.txt
ls -la $OUT
