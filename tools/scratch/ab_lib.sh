# A/B of two builds (tools/scratch/libA.so, libB.so) on the same box: the fp32 bench line, alternating
export MILLIEYE_TUNE_CACHE=/tmp/tune_ab.json
MILLIEYE_HIP_LIB=$PWD/tools/scratch/libA.so python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline --no-bf16-line > /dev/null 2>&1
for rep in 1 2 3; do
for v in A B; do
MILLIEYE_HIP_LIB=$PWD/tools/scratch/lib$v.so python bench.py --no-accuracy --no-batch-sweep --no-cpu-baseline --no-bf16-line ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], [s['ms'] for s in d['stages'][:2]])
"
done; done
