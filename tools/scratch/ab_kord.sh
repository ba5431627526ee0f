for rep in 1 2; do
for k in 1 2; do
MILLIEYE_KORD=$k MILLIEYE_TUNE_CACHE=/tmp/tune_k$k.json python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline --no-bf16-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kord=$k', d['value'], d['ms_per_step'], d['roofline']['frac'], [s['ms'] for s in d['stages'][:2]])
"
done; done
for k in 1 2; do
MILLIEYE_KORD=$k MILLIEYE_TUNE_CACHE=/tmp/tune_k$k.json BENCH_LAYERS=1 python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline --no-bf16-line --steps 10 2>&1 >/dev/null | grep "^.layer" | awk -v kb=$k '{print kb, $0}'
done
