#!/bin/bash
# A/B two builds of libcilrs_hip.so on one box: conv micro-benchmark + bench step, alternating.
PREV=$GRAFT_REPO_ROOT/cilrs-autonomous-driving-carla_amd/cilrs_mi355/libcilrs_hip_prev.so
for r in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export CILRS_LIB=$PREV; else unset CILRS_LIB; fi
    echo "== round $r $v"
    python tools/conv_bench.py 2>&1 | grep -E "c-1" | tr '\n' ' '; echo
    python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('step', d['value'], d['ms_per_step'])"
  done
done
