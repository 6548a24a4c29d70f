#!/bin/bash
# A/B on one box: wgrad with / without the pinned LDS-read interleave, three rounds each
for r in 1 2 3; do
  for pin in 0 1; do
    echo "== round $r CILRS_WGRAD_PIN=$pin"
    CILRS_WGRAD_PIN=$pin python tools/conv_bench.py --wgrad-only 2>&1 | grep -E "wgrad" | tr '\n' ' '
    echo
  done
done
