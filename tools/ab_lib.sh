#!/bin/bash
# A/B the product library against tools/bin/libcilrs_hip_exp.so (built with other -D flags) on ONE
# box: short bench line, three rounds round-robin; prints step time, frac and the conv families.
for rep in 1 2 3; do
  for lib in product exp; do
    if [ $lib = exp ]; then export CILRS_LIB=tools/bin/libcilrs_hip_exp.so; else unset CILRS_LIB; fi
    python bench.py --no-cpu-baseline --no-infer --no-loader 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fwd', d['kernels']['conv_fwd']['ms_per_step'], 'dgrad', d['kernels']['conv_dgrad']['ms_per_step'])"
  done
done
