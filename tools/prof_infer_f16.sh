#!/bin/bash
# rocprofv3 kernel stats of BASELINE configs[4]'s leg: five 64-frame streams through the fp16 trunk
# (tools/infer_streams.py: sequential, five lanes eager, five lanes from hipGraphs, one B=320 batch).
# Output: gpurun_out/prof_infer_f16/{kernel_stats.csv,streams.log}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_infer_f16
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/infer_streams.py > $OUT/streams.log 2> $OUT/streams.err
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/streams.log
