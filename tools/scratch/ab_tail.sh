for rep in 1 2; do
for tail in 0 1; do
MILLIEYE_TUNE_TAIL=$tail MILLIEYE_TUNE_CACHE=/tmp/tune_$tail.json python bench.py --no-accuracy --no-batch-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tail=$tail', d['value'], d['ms_per_step'], d['roofline']['frac'], [s['ms'] for s in d['stages'][:2]])
"
done; done
python - <<PY
import json
d=json.load(open('/tmp/tune_1.json'))
for k,v in d.items():
    if len(k.split(","))==9 and k.startswith("32,"): print(k,v)
PY
