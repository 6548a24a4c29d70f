#!/usr/bin/env python3
"""Find the periodic ~90 ms stall in the single-frame path: time each phase, try sync variants."""
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

def run(mode, n=120):
    slow = []
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        with torch.cuda.stream(pr.stream):
            if mode != "nocopy":
                pr.frames_dev.copy_(pr.frames_host, non_blocking=True)
                pr.speed_dev.copy_(pr.speed_host, non_blocking=True)
                pr.cmd_dev.copy_(pr.cmd_host, non_blocking=True)
            t1 = time.perf_counter()
            pr.eng.run_forward_u8(pr.frames_dev, pr.speed_dev, pr.cmd_dev, out=(pr.ctrl_dev, pr.spd_out_dev), graph=(mode != "eager"))
            t2 = time.perf_counter()
            if mode != "nocopy":
                pr.ctrl_host.copy_(pr.ctrl_dev, non_blocking=True)
            t3 = time.perf_counter()
            if mode == "spin":
                ev = torch.cuda.Event(); ev.record()
                while not ev.query():
                    pass
            else:
                pr.stream.synchronize()
            t4 = time.perf_counter()
        ts.append((t4 - t0) * 1e3)
        if t4 - t0 > 5e-3:
            slow.append((i, round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((t3 - t2) * 1e3, 2), round((t4 - t3) * 1e3, 2)))
    ts.sort()
    print(f"{mode:8s} median {ts[len(ts)//2]:.3f} ms, >5ms: {len(slow)}  (iter, h2d, launch, d2h, sync) {slow[:6]}")

for mode in ("graph", "eager", "nocopy", "spin", "graph"):
    run(mode)
import gc
gc.disable()
run("graph")
