#!/bin/bash
# per-kernel profile table of the bench with / without the system-scope fence on the timing events
for v in 0 1; do
  CILRS_PROF_NOFENCE=$v CILRS_LIB=tools/bin/libcilrs_hip_exp.so python bench.py --no-cpu-baseline --no-infer --no-loader 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CILRS_PROF_NOFENCE=$v', d['ms_per_step'], 'frac', d['roofline']['frac'], 'sum', d['kernels_sum_ms'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
done
