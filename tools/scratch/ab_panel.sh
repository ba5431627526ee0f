for rep in 1 2; do
for kb in 1000000 2400 1300; do
MILLIEYE_PANEL_KB=$kb MILLIEYE_TUNE_CACHE=/tmp/tune_p$kb.json python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline --no-bf16-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('panel_kb=$kb', d['value'], d['ms_per_step'], d['roofline']['frac'], [s['ms'] for s in d['stages'][:2]])
"
done; done
for kb in 1000000 2400 1300; do
MILLIEYE_PANEL_KB=$kb MILLIEYE_TUNE_CACHE=/tmp/tune_p$kb.json BENCH_LAYERS=1 python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline --no-bf16-line --steps 10 2>&1 >/dev/null | grep "^.layer" | awk -v kb=$kb '{print kb, $0}' | sed -n '27,50p'
done
