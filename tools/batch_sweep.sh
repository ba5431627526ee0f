#!/bin/bash
# frames/s at batch 1 / 8 / 32 (north_star's three batch sizes), fp32: bash tools/batch_sweep.sh > profiles/rNN_batch_sweep.txt
export MILLIEYE_TUNE_CACHE=/tmp/tune_sweep.json
for b in 1 8 32; do
  for wl in full detector; do
    python bench.py --no-cpu-baseline --no-bf16-line --workload $wl --batch $b --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32 %-8s batch %3d: %8.1f frames/s  %7.3f ms/step  conv %.1f TF (%.0f %% of the fp32 MFMA peak)' % ('$wl', $b, d['value'], d['ms_per_step'], d['roofline']['achieved'], 100*d['roofline']['frac']))"
  done
done
