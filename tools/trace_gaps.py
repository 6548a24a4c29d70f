#!/usr/bin/env python3
"""One production step from a rocprofv3 kernel trace: where the wall time goes -- union busy time,
idle gaps, and the time during which ONLY short (<12 us) kernels were running."""
import csv
import glob
import re
import sys
from collections import Counter

out = sys.argv[1]
import os
f = max(glob.glob(f"{out}/kt/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2] + 1:adam[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
ev = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
       (re.findall(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0])
      for r in step]
wall = max(e for _, e, _ in ev)
# sweep
pts = sorted([(s, 1, i) for i, (s, e, n) in enumerate(ev)] + [(e, -1, i) for i, (s, e, n) in enumerate(ev)])
active = set()
last = 0.0
idle = 0.0
small_only = 0.0
small_names = Counter()
for t, kind, i in pts:
    dt = t - last
    if dt > 0:
        if not active:
            idle += dt
        elif all(ev[j][1] - ev[j][0] < 12.0 for j in active):
            small_only += dt
            for j in active:
                small_names[ev[j][2]] += dt / len(active)
    last = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
print(f"step wall {wall:.1f} us, kernels {len(ev)}, idle {idle:.1f} us, only-short-kernels {small_only:.1f} us")
for n, v in small_names.most_common(14):
    print(f"   {n:42s} {v:8.1f} us")
