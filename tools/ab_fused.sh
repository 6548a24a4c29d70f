#!/bin/bash
# default step (backward + one Adam launch) against --fused-step (cilrs_net_backward_step), round-robin
for rep in 1 2 3; do
  for mode in "" "--fused-step"; do
    python bench.py --no-cpu-baseline --no-infer --no-loader $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode [$mode]', d['ms_per_step'], d['value'])"
  done
done
