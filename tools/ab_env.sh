#!/bin/bash
# A/B an experiment switch of the -DCILRS_EXPERIMENTS library on ONE box: ab_env.sh VAR v1 v2 ...
# runs the short bench line per value (twice round-robin) and prints step time, roofline.frac and
# the data-gradient family's serialised time.
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v CILRS_LIB=tools/bin/libcilrs_hip_exp.so python bench.py --no-cpu-baseline --no-infer --no-loader 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], 'frac', d['roofline']['frac'], 'dgrad', d['kernels']['conv_dgrad']['ms_per_step'], 'fwd', d['kernels']['conv_fwd']['ms_per_step'])"
  done
done
