export MILLIEYE_TUNE_VERBOSE=
python bench.py --workload detector --no-cpu-baseline --no-batch-sweep --no-accuracy --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('detector b8', d['value'], d['ms_per_step'], d['roofline']['frac'], [(s['stage'][:12], s['ms']) for s in d['stages'][:4]])
"
BENCH_LAYERS=1 python bench.py --workload detector --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 10 2>&1 >/dev/null | grep "^.layer" > gpurun_out/layers_det_b8.txt
BENCH_LAYERS=1 python bench.py --workload detector --batch 1 --no-cpu-baseline --no-batch-sweep --no-accuracy --no-bf16-line --steps 30 2>gpurun_out/layers_det_b1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('detector b1', d['value'], d['ms_per_step'], d['roofline']['frac'], [(s['stage'][:12], s['ms']) for s in d['stages'][:4]])
"
python bench.py --no-cpu-baseline --no-accuracy --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(d['batch_sweep'])[:1500])
"
