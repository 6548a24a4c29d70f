#!/bin/bash
# sweep the wgrad tile / split target on the conv micro-benchmark (tuning aid)
for bt in 0 64; do for tg in 256 512 768 1024 1536; do
  echo "### BT=$bt TARGET=$tg"
  CILRS_WGRAD_BT=$bt CILRS_WGRAD_TARGET=$tg python tools/conv_bench.py --iters 10 --wgrad-only 2>/dev/null | grep -E "^==|wgrad"
done; done
