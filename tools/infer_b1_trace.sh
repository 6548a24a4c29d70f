#!/bin/bash
# rocprofv3 kernel stats of the persistent single-frame path (csrc/infer_b1.hip): 300 control ticks
# through Predictor (one infer_b1_kernel launch per frame, zero-copy pinned I/O) and, for
# comparison, 100 ticks on the per-layer path.  Output: gpurun_out/infer_b1/{kernel_stats.csv,run.log}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/infer_b1
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/irun_b1.py <<PY
import sys, time
sys.path.insert(0, "$R/cilrs-autonomous-driving-carla_amd")
import numpy as np, torch
from cilrs_mi355 import CILRS
from cilrs_mi355.predict import Predictor
torch.manual_seed(0)
m = CILRS().cuda().eval()
frame = np.random.randint(0, 256, (88, 200, 3), dtype=np.uint8)
for name, kw, n in (("persistent", dict(), 300), ("per-layer launches (hipGraph)", dict(persistent=False), 100)):
    pr = Predictor(m, **kw)
    lat = []
    for i in range(n):
        t = time.perf_counter(); pr.predict_controls(frame, 25.0, i % 4); lat.append((time.perf_counter() - t) * 1e3)
    lat = sorted(lat[20:])
    print(f"{name}: median {lat[len(lat) // 2]:.4f} ms  p99 {lat[int(len(lat) * 0.99)]:.4f} ms  ({n} ticks)")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 /tmp/irun_b1.py > $OUT/run.log 2>&1
cat $OUT/run.log | grep -v amdgpu.ids
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-220
