#!/bin/bash
# Side-stream scheduling experiments: dy ring depth x side-stream priority (frames/s, Config A).
mkdir -p gpurun_out
for ring in 2 4 8; do
  for prio in 0 1; do
    echo "== CILRS_DY_RING=$ring CILRS_SIDE_PRIO=$prio"
    CILRS_DY_RING=$ring CILRS_SIDE_PRIO=$prio python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer --profile-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  done
done
