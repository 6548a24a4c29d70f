#!/bin/bash
# End-of-round evidence on ONE box: default bench line, serialised kernel stats, Config B at B=128 and
# B=120, the data-parallel path at world size 1.  Output: gpurun_out/<tag>/ (copy into profiles/).
T=${1:-final}; O=gpurun_out/$T; mkdir -p $O
python bench.py > $O/bench_final.log 2> $O/bench_final.err || exit 1
cut -c1-330 $O/bench_final.log
bash tools/prof_serial.sh > $O/prof_serial.out 2>&1
python bench.py --config B --no-cpu-baseline --no-infer --no-loader > $O/bench_cfgB.log 2>/dev/null
python bench.py --config B --batch 120 --no-cpu-baseline --no-infer --no-loader > $O/bench_cfgB120.log 2>/dev/null
python bench.py --force-dp --no-cpu-baseline --no-infer --no-loader > $O/bench_dp1.log 2> $O/bench_dp1.err
for f in cfgB cfgB120 dp1; do python -c "
import json
d=[json.loads(l) for l in open('$O/bench_$f.log') if l.startswith('{')][-1]; print('$f', d['value'], d['ms_per_step'])"; done
