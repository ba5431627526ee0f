for rep in 1 2 3; do
python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], [s['ms'] for s in d['stages'][:2]], d['bf16_storage_mode']['value'], d['bf16_storage_mode']['roofline']['frac'])
"
done
