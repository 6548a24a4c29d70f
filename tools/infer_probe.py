#!/usr/bin/env python3
"""Where does single-frame latency go?  Device-only graph replay time vs end-to-end predict."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
import numpy as np
import torch
from cilrs_mi355 import CILRS
from cilrs_mi355.predict import Predictor

torch.manual_seed(0)
m = CILRS().cuda().eval()
pr = Predictor(m)
frame = np.random.randint(0, 256, (88, 200, 3), dtype=np.uint8)
for _ in range(20):
    pr.predict_controls(frame, 25.0, 0)
lat = []
for _ in range(300):
    t = time.perf_counter(); pr.predict_controls(frame, 25.0, 0); lat.append((time.perf_counter() - t) * 1e3)
lat.sort(); print(f"end-to-end predict_controls: median {lat[150]:.3f} ms  p10 {lat[30]:.3f}  p90 {lat[270]:.3f}")
# device-only: graph replays back to back on the predictor's stream
eng = pr.eng
with torch.cuda.stream(pr.stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        eng.run_forward_u8(pr.frames_dev, pr.speed_dev, pr.cmd_dev, out=(pr.ctrl_dev, pr.spd_out_dev), graph=True)
    e1.record(); pr.stream.synchronize()
print(f"device, graph replay back-to-back: {e0.elapsed_time(e1) / 200:.3f} ms per forward")
with torch.cuda.stream(pr.stream):
    e0.record()
    for _ in range(200):
        eng.run_forward_u8(pr.frames_dev, pr.speed_dev, pr.cmd_dev, out=(pr.ctrl_dev, pr.spd_out_dev), graph=False)
    e1.record(); pr.stream.synchronize()
print(f"device, eager launches back-to-back: {e0.elapsed_time(e1) / 200:.3f} ms per forward")
t = time.perf_counter()
for _ in range(200):
    with torch.cuda.stream(pr.stream):
        eng.run_forward_u8(pr.frames_dev, pr.speed_dev, pr.cmd_dev, out=(pr.ctrl_dev, pr.spd_out_dev), graph=True)
        pr.stream.synchronize()
print(f"graph launch + sync (no copies): {(time.perf_counter() - t) / 200 * 1e3:.3f} ms")
os.environ["CILRS_OVERLAP"] = "1"
