#!/bin/bash
# The data-parallel step at world size 1 (--force-dp: segment-wise backward, three all-reduce buckets,
# one Adam launch) against the single-process step (backward + one Adam launch) and its fused form (--fused-step), round-robin on ONE box.
for rep in 1 2; do
  for mode in "" "--fused-step" "--force-dp"; do
    python bench.py --no-cpu-baseline --no-infer --no-loader $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode [$mode]', d['ms_per_step'], d['value'])"
  done
done
