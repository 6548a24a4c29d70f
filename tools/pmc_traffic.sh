#!/bin/bash
# HBM traffic of the dominant conv kernels via rocprofv3 PMC (separate passes: FETCH_SIZE needs 3
# of the 4 TCC slots, WRITE_SIZE 2).  Run on the GPU box; writes gpurun_out/traffic/.
# usage: tools/pmc_traffic.sh <commit the tree was built from> (recorded in traffic.json)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer --no-loader --profile-steps 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
python3 $R/tools/pmc_traffic_summary.py $OUT ${1:-unknown} > $OUT/summary.log 2>&1
cat $OUT/summary.log
