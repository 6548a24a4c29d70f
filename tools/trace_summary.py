#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of bench.py: per-kernel totals for the LAST training step
and the device idle time inside it (gaps between kernels, accounting for concurrent streams)."""
import csv
import re
import glob
import sys

out = sys.argv[1]
f = glob.glob(f"{out}/kt/*/*_kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps are delimited by the adam kernel
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
if len(adam) < 3:
    print("not enough steps", len(adam)); sys.exit(0)
lo, hi = adam[-2] + 1, adam[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for s, e, n in step:
    m = re.search(r"(\w+_kernel)\b(<[^>]*>)?", n)
    k = (m.group(1) + (m.group(2) or "")) if m else n[:48]
    a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
print(f"step wall {1e-3*(t1-t0):.1f} us, device busy (union) {1e-3*busy:.1f} us, idle {1e-3*(t1-t0-busy):.1f} us, "
      f"kernels {len(step)}, sum of kernel times {1e-3*sum(e-s for s,e,_ in step):.1f} us")
for k, (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:50s} x{c:4d} {1e-3*ns:9.1f} us")
