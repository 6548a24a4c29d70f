#!/usr/bin/env python3
"""Single-frame latency of the persistent launch (csrc/infer_b1.hip) next to the per-layer launch
path: end to end (predict_controls), device only (back-to-back launches), and -- with
CILRS_B1_STAMPS=1 -- block 0's per-stage clock inside the launch."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import numpy as np
import torch
from cilrs_mi355 import CILRS, _lib as L
from cilrs_mi355.predict import Predictor

torch.manual_seed(0)
m = CILRS().cuda().eval()
frame = np.random.randint(0, 256, (88, 200, 3), dtype=np.uint8)
for name, kw in (("per-layer launches", dict(persistent=False)), ("persistent launch", dict())):
    pr = Predictor(m, **kw)
    for _ in range(50):
        pr.predict_controls(frame, 25.0, 0)
    lat = []
    for _ in range(1000):
        t = time.perf_counter(); pr.predict_controls(frame, 25.0, 0); lat.append((time.perf_counter() - t) * 1e3)
    lat.sort()
    eng = pr.eng
    with torch.cuda.stream(pr.stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            eng.run_forward_u8(pr.frames_dev, pr.speed_dev, pr.cmd_dev, out=(pr.ctrl_dev, pr.spd_out_dev),
                               persistent=pr.persistent)
        e1.record(); pr.stream.synchronize()
    print(f"{name:20s}: predict_controls median {lat[500]:.4f} ms  p10 {lat[100]:.4f}  p99 {lat[990]:.4f};"
          f"  device, back to back {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per frame")
if os.environ.get("CILRS_B1_STAMPS"):
    pl = pr.eng.plan(1, 88, 200)
    n = L.lib().cilrs_net_b1_stages(pl.handle)
    st, wk = (C.c_float * n)(), (C.c_float * n)()
    pr.predict_controls(frame, 25.0, 0)
    L.check(L.lib().cilrs_net_b1_stage_us(pl.handle, C.byref(pl.bufs), st, wk, n))
    names = ["pre", "stem", "pool"] + [f"blk{b}.{c}" for b in range(16) for c in ("conv1", "conv2")] + ["head1", "head2", "head3"]
    print("stage        start_us  block0_work_us  stage_us")
    for i in range(n):
        nxt = st[i + 1] if i + 1 < n else st[i] + wk[i]
        print(f"{names[i]:12s} {st[i]:8.2f} {wk[i]:10.2f} {nxt - st[i]:10.2f}")
