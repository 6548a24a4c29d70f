#!/usr/bin/env python3
"""Latency distribution of Predictor.predict_controls (periodic outliers?), gc on / off."""
import os, sys, time, gc
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
for label, gcoff, graph in (("gc on graph", False, True), ("gc on eager", False, False), ("gc off graph", True, True)):
    if gcoff:
        gc.disable()
    pr.use_graph = graph
    tot, slow = [], []
    for i in range(300):
        t0 = time.perf_counter()
        pr.predict_controls(frame, 25.0, 0)
        dt = (time.perf_counter() - t0) * 1e3
        tot.append(dt)
        if dt > 5:
            slow.append((i, round(dt, 1)))
    tot.sort()
    print(label, "median", round(tot[150], 3), "p99", round(tot[297], 3), "slow(>5ms)", len(slow), slow[:8])
