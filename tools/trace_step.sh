#!/bin/bash
# kernel timeline of a few training steps (rocprofv3 --kernel-trace --stats), summary to gpurun_out/trace/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-loader --two-call-step --profile-steps 0 > $OUT/bench.log 2>&1
python3 $R/tools/trace_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
