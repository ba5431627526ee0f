R=$PWD; export TMPDIR=/tmp; export MILLIEYE_TUNE_CACHE=/tmp/tune_pl.json
python bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 3 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"roofline\"][\"frac\"], d[\"bf16_storage_mode\"][\"value\"])"
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 3 --warmup 1 --prewarm-seconds 0.3"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/plf -o f -- $CMD > /tmp/plf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/plw -o w -- $CMD > /tmp/plw.log 2>&1
python $R/tools/pmc_layers.py /tmp/plf/f_results.db /tmp/plw/w_results.db 32 416 f32 > $R/gpurun_out/layer_traffic_f32.txt 2>&1
CMD="$CMD --dtype bf16"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/plf2 -o f -- $CMD > /tmp/plf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/plw2 -o w -- $CMD > /tmp/plw.log 2>&1
python $R/tools/pmc_layers.py /tmp/plf2/f_results.db /tmp/plw2/w_results.db 32 416 bf16 > $R/gpurun_out/layer_traffic_bf16.txt 2>&1
tail -n 3 $R/gpurun_out/layer_traffic_f32.txt; tail -n 3 $R/gpurun_out/layer_traffic_bf16.txt
