#!/bin/bash
# rocprofv3 kernel stats of the ResNet-50 variant's train step in the fp32 and the bf16 mode
# (tools/bf16_train_probe.py runs both): per-kernel averages of the 16-bit training kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bf16
rm -rf $OUT; mkdir -p $OUT
export CILRS_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/bf16_train_probe.py resnet50 64 > $OUT/probe.log 2> $OUT/probe.err
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-160
