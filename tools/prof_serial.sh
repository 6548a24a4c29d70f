#!/bin/bash
# Serialised (CILRS_OVERLAP=0) rocprofv3 kernel stats of the bench: per-kernel averages comparable
# with bench.py's hipEvent brackets.  Output: gpurun_out/prof_serial/{kernel_stats.csv,bench_line.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_serial
rm -rf $OUT; mkdir -p $OUT
export CILRS_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-loader > $OUT/bench_line.json 2> $OUT/bench.err
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
head -40 $OUT/kernel_stats.csv | cut -c1-200
