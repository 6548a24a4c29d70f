#!/bin/bash
# kernel timeline of the data-parallel step at world size 1 (--force-dp): where do the three
# all-reduce launches sit, how long do they run, what does the device do meanwhile
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_dp
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-loader --force-dp --profile-steps 0 > $OUT/bench.log 2>&1
python3 - <<P > $OUT/summary.txt 2>&1
import csv, glob, os
f = sorted(glob.glob("$OUT/kt/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 10]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0 = step[0][0]
busy, cs, ce = 0, None, None
for s, e, _, _ in step:
    if ce is None or s > ce:
        if ce is not None: busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
wall = max(r[1] for r in step) - t0
print(f"step wall {wall/1e3:.1f} us, device busy (union) {busy/1e3:.1f} us, kernels {len(step)}, queues {sorted(set(r[3] for r in step))}")
for s, e, n, q in step:
    if "ccl" in n.lower() or "allreduce" in n.lower() or "AllReduce" in n or "adam" in n:
        print(f"  t={(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us  queue {q}  {n[:90]}")
P
cat $OUT/summary.txt
