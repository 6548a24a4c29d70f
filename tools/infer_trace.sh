#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/itrace
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/irun.py <<PY
import os, sys, time
sys.path.insert(0, "$R/cilrs-autonomous-driving-carla_amd")
import numpy as np, torch
from cilrs_mi355 import CILRS
from cilrs_mi355.predict import Predictor
torch.manual_seed(0)
m = CILRS().cuda().eval()
pr = Predictor(m, use_graph=False)
frame = np.random.randint(0, 256, (88, 200, 3), dtype=np.uint8)
lat=[]
for i in range(60):
    t=time.perf_counter(); pr.predict_controls(frame, 25.0, 0); lat.append((time.perf_counter()-t)*1e3)
print("eager lat ms:", " ".join(f"{x:.2f}" for x in lat[20:60]))
pr.use_graph = True
lat=[]
for i in range(60):
    t=time.perf_counter(); pr.predict_controls(frame, 25.0, 0); lat.append((time.perf_counter()-t)*1e3)
print("graph lat ms:", " ".join(f"{x:.2f}" for x in lat[20:60]))
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 /tmp/irun.py > $OUT/run.log 2>&1
grep "lat ms" $OUT/run.log
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/kt/*/*_kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Z"]) for r in csv.DictReader(open(f))))
# last eager forward = between the 39th and 40th-from-... find u8 transform kernels
idx = [i for i, r in enumerate(rows) if "u8hwc_to_nhwc4" in r[2]]
lo, hi = idx[50], idx[51]
seg = rows[lo:hi]
print("kernels in one forward:", len(seg), "span us", (seg[-1][1] - seg[0][0]) / 1e3, "sum kernel us", sum(e - s for s, e, *_ in seg) / 1e3)
prev_end = None
for s, e, n, gx, gz in seg:
    m = re.search(r"(\w+_kernel)\b(<[^>]*>)?", n)
    k = (m.group(1) + (m.group(2) or "")) if m else n[:40]
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{k[:52]:52s} grid={int(gx)//256:5d}x{gz} dur={(e - s) / 1e3:6.1f}us gap={gap:6.1f}")
    prev_end = max(prev_end or 0, e)
PY
